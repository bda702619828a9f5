#!/usr/bin/env python3
"""Randomised differential run of the speculative index (forced on) against the CPU oracle: many small
plans of mixed packets — encoder-made at random quality/noise, with unchanged (0xFF) blocks, cut short at
lengths around the walker chunk size, random bytes, runs of tiny blocks.  Not part of the test suite
(minutes); run on the GPU box: python tools/stress_spec.py [seconds] [seed] [MI_RTJ_SPEC mode]."""
import os, sys, time

os.environ["MI_RTJ_SPEC"] = sys.argv[3] if len(sys.argv) > 3 else "1"  # 1: short lead, 3: long lead, 2: with the policy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rtjlib as R
from pkg import P

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = P.MiRtj()
K = 2048


def hdr(w, h, Q, n):
    t = 12 + n
    return np.array([t & 255, (t >> 8) & 255, (t >> 16) & 255, (t >> 24) & 255, 12, 0, w & 255, w >> 8, h & 255, h >> 8, Q, 0], np.uint8)


def make_packet():
    w, h = [(64, 48), (160, 128), (320, 240), (640, 368), (1920, 1088), (1024, 16)][int(rng.integers(0, 6))]
    Q = int(rng.choice([1, 20, 64, 100, 128, 170, 171, 200, 230, 255]))
    kind = int(rng.integers(0, 8))
    if kind <= 3:
        amp = int(rng.choice([0, 2, 8, 12, 18, 30]))
        if kind == 3:
            enc = R.OracleEncoder(w, h, Q, key_rate=int(rng.integers(2, 9)), lmask=int(rng.integers(0, 6)), cmask=int(rng.integers(0, 6)))
            for i in range(int(rng.integers(1, 4))):
                p = enc.encode(R.synth_frame(w, h, i // 2, seed=int(rng.integers(0, 99)), amp=amp))
        else:
            p = R.OracleEncoder(w, h, Q).encode(R.synth_frame(w, h, int(rng.integers(0, 50)), seed=int(rng.integers(0, 99)), amp=amp))
        if rng.random() < 0.3:  # cut short, often near a multiple of the walker chunk
            n = p.size - 12
            cut = int(rng.integers(0, n + 1)) if rng.random() < 0.5 else max(0, min(n, int(rng.integers(1, 1 + max(1, n // K))) * K + int(rng.integers(-3, 4))))
            p = np.concatenate([hdr(w, h, Q, cut), p[12:12 + cut]])
        return p
    n = int(rng.integers(0, (w // 16) * (h // 16) * 6 * 40 + 50))
    if kind == 4:
        body = rng.integers(0, 256, n, dtype=np.uint8)
    elif kind == 5:
        body = rng.choice(np.array([0xFF, 0x7E, 0x10, 0x7F, 0x40], np.uint8), n)
    elif kind == 6:  # two-byte blocks: DC + full run
        body = np.tile(np.array([0x33, 0x7E], np.uint8), n // 2 + 1)[:n]
    else:
        body = rng.integers(0, 6, n, dtype=np.uint8)
    return np.concatenate([hdr(w, h, Q, n), body])


t0, plans, packets, proven_total = time.time(), 0, 0, 0
last_note = t0
while time.time() - t0 < budget:
    pkts = [make_packet() for _ in range(int(rng.integers(1, 24)))]
    d_stream, po, pl, hdrs = dev.upload_packets(pkts, align=int(rng.choice([1, 4, 64])))
    sizes = [(int(p[6]) | int(p[7]) << 8) * (int(p[8]) | int(p[9]) << 8) * 3 // 2 for p in pkts]
    oo = np.concatenate([[0], np.cumsum([(s + 255) // 256 * 256 for s in sizes])]).astype(np.uint64)
    d_out = dev.alloc(int(oo[-1]))
    dev.memset(d_out, 0x5A, int(oo[-1]))
    q_before = dev.state()[2]
    plan = dev.plan(hdrs, po, pl, oo[:-1].copy())
    # with the policy (mode 2) a plan is decoded four times: its state may go short lead -> long lead -> paused,
    # and every decode must give the same pictures
    for rep in range(4 if os.environ["MI_RTJ_SPEC"] == "2" else 1):
        if rep:
            for i, p in enumerate(pkts):
                assert np.array_equal(dev.d2h(d_out, sizes[i], offset=int(oo[i])), first[i]), ("repeat", plans, rep, i)
            dev.memset(d_out, 0x5A, int(oo[-1]))
        plan.decode(d_stream, d_out)
        dev.sync()
        if rep == 0:
            first = [dev.d2h(d_out, sizes[i], offset=int(oo[i])) for i in range(len(pkts))]
    proven_total += plan.spec_stats()[0]
    idx = plan.read_index()
    dec = R.OracleDecoder()
    if q_before:
        dec.decode(R.OracleEncoder(16, 16, q_before).encode(R.synth_frame(16, 16, 0)), np.zeros(384, np.uint8))
    k = 0
    for i, p in enumerate(pkts):
        want = np.full(sizes[i], 0x5A, np.uint8)
        dec.decode(p, want)
        got = dev.d2h(d_out, sizes[i], offset=int(oo[i]))
        nb = (sizes[i] * 2 // 3) // 64 * 6 // 4 + 1  # blocks + 1
        assert np.array_equal(got, want), ("planes", plans, i, p[:12].tolist(), p.size)
        k += nb
    plan.close()
    dev.free(d_stream)
    dev.free(d_out)
    plans += 1
    packets += len(pkts)
    if time.time() - last_note > 60:  # gpurun takes a silent command for hung
        last_note = time.time()
        print(f"... {plans} plans, {packets} packets, {last_note - t0:.0f} s", flush=True)
print(f"stress ok: {plans} plans, {packets} packets, {proven_total} proven by the speculative index, {time.time() - t0:.0f} s")
