"""Builds libmi_rtjpeg.so (HIP kernels + C ABI) for gfx950, in tree.

    python gmerlin-avdecoder_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmi_rtjpeg.so")
SOURCES = ["mi_rtjpeg.hip", "rtj_tables.cpp"]
import glob


def _deps():
    d = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + \
        glob.glob(os.path.join(CSRC, "*.cpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return d
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force=False, verbose=False, out=None):
    if not force and not stale() and out is None:
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    extra = os.environ.get("MI_RTJ_CFLAGS", "").split()
    # compiled in a scratch directory with the device assembly kept: EVERY library built here (the product, the test
    # variant, A/B builds with other MI_RTJ_CFLAGS) has its k_decode checked for the one thing the compiler does not
    # know — the hand-issued look-ahead loads and their counted wait (tools/check_async_loads.py) — and a finding
    # fails the build
    tmp = tempfile.mkdtemp(prefix="mirtj_build_")
    try:
        so = os.path.join(tmp, "lib.so")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", "-save-temps=obj"] + extra + ["-o", so] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        asm = os.path.join(tmp, "mi_rtjpeg-hip-amdgcn-amd-amdhsa-gfx950.s")
        chk = os.path.join(HERE, "..", "tools", "check_async_loads.py")
        if os.path.exists(chk):  # (no way around it: round 3's MI_RTJ_SKIP_ASYNC_CHECK is gone)
            c = subprocess.run([sys.executable, chk, asm], capture_output=True, text=True)
            if c.returncode != 0:
                raise RuntimeError("k_decode's hand-issued loads are not safe in this build:\n" + c.stdout + c.stderr)
        shutil.move(so, out or LIB)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out or LIB


DV_LIB = os.path.join(LIBDIR, "libmi_dv.so")
DV_SOURCES = ["mi_dv.hip", "dv_tables.cpp"]


def build_dv(force=False, verbose=False):
    """libmi_dv.so: the DV25 525/60 decoder (include/mi_dv.h), gfx950."""
    deps = [os.path.join(CSRC, f) for f in DV_SOURCES + ["dv_common.h", "dv_decode_kernels.h"]] + \
           [os.path.join(HERE, "..", "include", "mi_dv.h")]
    if not force and os.path.exists(DV_LIB) and all(os.path.getmtime(f) <= os.path.getmtime(DV_LIB) for f in deps):
        return DV_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"] + \
        os.environ.get("MI_DV_CFLAGS", "").split() + ["-o", DV_LIB] + [os.path.join(CSRC, s) for s in DV_SOURCES]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return DV_LIB


TEST_VARIANT = os.path.join(LIBDIR, "libmi_rtjpeg_generic_paths.so")


def build_test_variant(force=False):
    """The same library compiled with -DMIRTJ_TEST_GENERIC_PATHS: the kernels always take the paths
    that real tables and whole packets rarely reach (run-time raw-byte count in the parse loop, masked
    loads near a packet's end).  Loaded only by tests/test_gpu_variant_paths.py."""
    if not force and os.path.exists(TEST_VARIANT) and \
            all(os.path.getmtime(f) <= os.path.getmtime(TEST_VARIANT) for f in _deps()):
        return TEST_VARIANT
    old = os.environ.get("MI_RTJ_CFLAGS")
    os.environ["MI_RTJ_CFLAGS"] = ((old + " ") if old else "") + "-DMIRTJ_TEST_GENERIC_PATHS"
    try:
        return build(force=True, out=TEST_VARIANT)
    finally:
        if old is None:
            del os.environ["MI_RTJ_CFLAGS"]
        else:
            os.environ["MI_RTJ_CFLAGS"] = old


EXP_LIB = os.path.join(LIBDIR, "libmi_rtjpeg_exp.so")


def build_experiments(force=False):
    """The same library with -DMIRTJ_EXPERIMENTS: the measurement switches that leave work out or run forms that were
    measured slower (exp_env() in mi_rtjpeg.hip) exist in this build only.  For tools/ (MI_RTJ_LIB=...), never shipped
    as the product and not built by __graft_entry__.build()."""
    if not force and os.path.exists(EXP_LIB) and all(os.path.getmtime(f) <= os.path.getmtime(EXP_LIB) for f in _deps()):
        return EXP_LIB
    old = os.environ.get("MI_RTJ_CFLAGS")
    os.environ["MI_RTJ_CFLAGS"] = ((old + " ") if old else "") + "-DMIRTJ_EXPERIMENTS"
    try:
        return build(force=True, out=EXP_LIB)
    finally:
        if old is None:
            del os.environ["MI_RTJ_CFLAGS"]
        else:
            os.environ["MI_RTJ_CFLAGS"] = old


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_dv(force="--force" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build_experiments(force=True))
