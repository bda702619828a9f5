"""ctypes view of oracle/libdv_oracle.so (oracle/dv_oracle.h): the CHECKER of the DV path.  Tests and bench.py's
cpu_baseline leg only; the product never loads it."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(os.environ.get("MI_SAN_LIBDIR") or os.path.join(ROOT, "oracle"), "libdv_oracle.so")
FRAME_BYTES, PICTURE_BYTES, W, H, CW = 120000, 720 * 480 * 3 // 2, 720, 480, 180
u8p = C.POINTER(C.c_uint8)
_L = None


def lib():
    global _L
    if _L is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
        _L = C.CDLL(LIB)
        _L.dvo_encode_frame.argtypes = [u8p, u8p, C.c_int]
        _L.dvo_decode_frame.argtypes = [u8p, u8p]
        _L.dvo_qbase.argtypes = [C.c_int, C.POINTER(C.c_int32)]
        _L.dvo_scan.argtypes = [C.c_int, u8p]
        _L.dvo_shift.argtypes = [C.c_int] * 3
        _L.dvo_block_pixels.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int16), u8p]
        _L.dvo_mb_place.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)] * 2
        _L.dvo_vlc_lookup.argtypes = [C.c_uint32] + [C.POINTER(C.c_int)] * 3
        _L.dvo_synth.argtypes = [u8p, C.c_int, C.c_uint32, C.c_int]
    return _L


def p8(a):
    return a.ctypes.data_as(u8p)


def synth(n, seed=1, amp=8):
    pic = np.zeros(PICTURE_BYTES, np.uint8)
    lib().dvo_synth(p8(pic), n, seed, amp)
    return pic


def encode(pic, flags=3):
    dif = np.zeros(FRAME_BYTES, np.uint8)
    lib().dvo_encode_frame(p8(np.ascontiguousarray(pic)), p8(dif), flags)
    return dif


def decode(dif):
    pic = np.zeros(PICTURE_BYTES, np.uint8)
    lib().dvo_decode_frame(p8(np.ascontiguousarray(dif)), p8(pic))
    return pic


def vlc(bits16):
    ln, run, lv = C.c_int(), C.c_int(), C.c_int()
    eob = lib().dvo_vlc_lookup(bits16, C.byref(ln), C.byref(run), C.byref(lv))
    return ln.value, run.value, lv.value, bool(eob)


def video_block_offset(seq, v):
    return (seq * 150 + 7 + v + v // 15) * 80


AREA_OFF = (4, 18, 32, 46, 60, 70)
