"""The host C of the path under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU sanitizers do not exist
on this pool): `make san` builds the QuickTime reader/writer (it parses untrusted files), the DV DIF handler and the
oracle with -fsanitize=address,undefined, and the suites that drive them — the damaged-movie cases, the DV audio
de-shuffle, the oracle against the reference build and the golden vectors — run again on those libraries in a child
python with the sanitizer runtime preloaded.  The wrapper and the harness are compiled with the same flags (object
code only: linking them needs libmi_rtjpeg.so, whose HIP runtime is not a sanitizer target)."""
import os
import subprocess
import sys

import pytest

from pkg import ROOT

SANDIR = os.path.join(ROOT, "gmerlin-avdecoder_amd", "lib", "san")


def runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(runtime("libasan.so") is None, reason="no sanitizer runtime in this toolchain")
def test_host_c_suites_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "gmerlin-avdecoder_amd", "csrc"), "san"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for f in ("libmi_qtrtj.so", "libmi_dvframe.so", "librtj_oracle.so", "libdv_oracle.so", "plugin_harness.o", "video_rtjpeg_mi355x.o",
              "video_dv_mi355x.o"):
        assert os.path.exists(os.path.join(SANDIR, f)), f
    env = dict(os.environ, MI_SAN_LIBDIR=SANDIR, LD_PRELOAD=runtime("libasan.so") + ":" + runtime("libubsan.so"),
               # python itself leaks by design; an error must kill the test process (abort), not just print
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # (the DV statement decodes arbitrary bytes in its golden test: every bit reader and table of it under the sanitizers)
    suites = ["tests/test_qt_rtj0_host.py", "tests/test_dvframe_host.py", "tests/test_oracle_golden.py",
              "tests/test_oracle_vs_reference.py", "tests/test_dv_oracle.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + suites,
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
