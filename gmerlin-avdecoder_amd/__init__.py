"""MI355X-native RTjpeg frame-decode path for gmerlin-avdecoder.

The product is libmi_rtjpeg.so (csrc/, C ABI in include/mi_rtjpeg.h) plus the C plugin wrapper
that binds it into the reference's bgav_video_decoder_t table.  This Python package is only the
ctypes view of that ABI used by tests/ and bench.py."""
from .binding import (MiRtj, MiRtjError, Plan, device_count, get_tables, lib_path, load)  # noqa: F401
