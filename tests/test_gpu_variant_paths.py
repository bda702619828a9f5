"""The kernels specialise for what real streams look like (raw-byte counts 0/4/8/9 known at compile
time, unmasked loads when a wave reads inside its packet).  A second build of the same library with
-DMIRTJ_TEST_GENERIC_PATHS always takes the general paths instead; this test runs the parity checks on it, in a
child process (one library per process)."""
import os
import subprocess
import sys

import pytest

from pkg import P

pytestmark = pytest.mark.gpu

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import rtjlib as R
from pkg import P
dev = P.MiRtj()
rng = np.random.default_rng(7)
pkts = []
for (w, h, Q, amp) in [(320, 240, 255, 8), (320, 240, 200, 30), (176, 144, 150, 20), (64, 48, 60, 64), (1920, 1088, 255, 8),
                       (320, 240, 255, 0)]:
    enc = R.OracleEncoder(w, h, Q)
    pkts += [enc.encode(R.synth_frame(w, h, i, seed=3, amp=amp)) for i in range(2)]
inter = R.OracleEncoder(320, 240, 200, key_rate=5, lmask=2, cmask=2)
stream = [inter.encode(R.synth_frame(320, 240, n // 3, seed=9, amp=2)) for n in range(6)]
for n in (0, 7000, 40000):  # arbitrary payloads, also far too short ones
    body = rng.integers(0, 256, n, dtype=np.uint8)
    total = 12 + n
    hdr = np.array([total & 255, (total >> 8) & 255, (total >> 16) & 255, 0, 12, 0, 160, 0, 64, 0, 129, 0], np.uint8)
    pkts.append(np.concatenate([hdr, body]))
d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=1)
sizes = [(int(p[6]) | int(p[7]) << 8) * (int(p[8]) | int(p[9]) << 8) * 3 // 2 for p in pkts]
oo = np.concatenate([[0], np.cumsum([(s + 255) // 256 * 256 for s in sizes])]).astype(np.uint64)
d_out = dev.alloc(int(oo[-1]))
dev.memset(d_out, 0x44, int(oo[-1]))
plan = dev.plan(hdrs, po, pl, oo[:-1].copy())
plan.decode(d_stream, d_out)
dev.sync()
dec = R.OracleDecoder()
for i, p in enumerate(pkts):
    want = np.full(sizes[i], 0x44, np.uint8)
    dec.decode(p, want)
    got = dev.d2h(d_out, sizes[i], offset=int(oo[i]))
    assert np.array_equal(got, want), ("batch", i)
one, od = P.MiRtj(), R.OracleDecoder()
got = np.zeros(320 * 240 * 3 // 2, np.uint8)
want = got.copy()
for n, p in enumerate(stream):  # in-order stream with unchanged blocks through the one-packet path
    one.decode(p, got)
    od.decode(p, want)
    assert np.array_equal(got, want), ("stream", n)
print("variant ok", len(pkts) + len(stream))
"""


def test_general_kernel_paths_match_the_oracle(tmp_path):
    bld = __import__("importlib").import_module("gmerlin-avdecoder_amd.build")
    lib = bld.build_test_variant()  # built by __graft_entry__.build(); rebuilt here only if missing or stale
    tests = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MI_RTJ_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CHILD, tests], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("which", ["product", "general-paths"])
def test_waves_that_take_all_three_parts_match_the_oracle(which):
    """Large batches run k_decode with one wave per slot, taking the upper luma, lower luma and chroma blocks of each of
    its groups in turn (span 3; the stream bytes then cross the fabric once).  MI_RTJ_ROTATE=1 asks for that
    arrangement whatever the batch size: the same packets, mixed geometries and tables in one plan, the in-order
    stream with unchanged blocks, arbitrary and truncated payloads."""
    tests = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MI_RTJ_ROTATE="1")
    if which != "product":
        bld = __import__("importlib").import_module("gmerlin-avdecoder_amd.build")
        env["MI_RTJ_LIB"] = bld.build_test_variant()
    r = subprocess.run([sys.executable, "-c", CHILD, tests], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
