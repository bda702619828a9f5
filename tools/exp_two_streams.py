#!/usr/bin/env python3
"""Experiment: does splitting the batch over two instances (two HIP streams) overlap the latency-bound
index kernels of one half with the ALU-bound kernels of the other?"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("gmerlin-avdecoder_amd")
w, h, Q, n = 1920, 1088, 255, int(os.environ.get("N", "1024"))
fsz = w * h * 3 // 2

def setup(dev, first, cnt):
    d_fr = dev.synth(w, h, first, cnt)
    d_st, po, pl = dev.encode(w, h, Q, cnt, d_fr)
    dev.sync(); dev.free(d_fr)
    hdr = dev.d2h(d_st, 12, offset=int(po[0]))
    d_out = dev.alloc(fsz * cnt)
    plan = dev.plan(np.tile(hdr, (cnt, 1)), po, pl, np.arange(cnt, dtype=np.uint64) * np.uint64(fsz))
    return d_st, d_out, plan

def run(parts, steps=20):
    devs = [P.MiRtj() for _ in range(parts)]
    sets = [setup(d, i * (n // parts), n // parts) for i, d in enumerate(devs)]
    for _ in range(3):
        for (st, out, plan) in sets: plan.decode(st, out)
    for d in devs: d.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for (st, out, plan) in sets: plan.decode(st, out)
    for d in devs: d.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"{parts} stream(s): {dt*1e3:.4f} ms per {n} frames -> {n/dt:.0f} fps")

for parts in (1, 2, 4, 1, 2, 4):
    run(parts)
