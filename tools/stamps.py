#!/usr/bin/env python3
"""Diagnostic: where a k_decode wave's time goes.  Needs a library built with -DMIRTJ_STAMPS (MI_RTJ_LIB=...);
prints shader cycles per section per wave.  Usage on the GPU box: MI_RTJ_LIB=... python tools/stamps.py [bench args]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("gmerlin-avdecoder_amd")
amp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1024
dev = P.MiRtj(0)
w, h, Q = 1920, 1088, 255
d_fr = dev.synth(w, h, 0, n, seed=12345, amp=amp)
d_st, po, pl = dev.encode(w, h, Q, n, d_fr)
dev.sync(); dev.free(d_fr)
hdr0 = dev.d2h(d_st, 12, offset=int(po[0]))
fsz = w * h * 3 // 2
plan = dev.plan(np.tile(hdr0, (n, 1)), po, pl, np.arange(n, dtype=np.uint64) * np.uint64(fsz))
d_out = dev.alloc(fsz * n)
L = dev.L
out = (C.c_ulonglong * 16)()
for _ in range(3):
    plan.decode(d_st, d_out)
dev.sync()
L.mi_rtj_debug_stamps(out)
for _ in range(4):
    plan.decode(d_st, d_out)
dev.sync()
L.mi_rtj_debug_stamps(out)
v = list(out)
waves = v[7]
names = ["prologue", "classify+parse", "prefetch+lo test", "transform+stores", "counted wait", "coords+reads+range test", "-"]
tot = sum(v[:7]) + sum(v[8:15])
print(f"amp {amp}: waves {waves}, cycles per wave {tot / waves:.0f}")
for base, what in ((0, "luma"), (8, "chroma")):
    for nme, x in zip(names, v[base:base + 7]):
        print(f"  {what:6s} {nme:26s} {x / waves:10.0f} cycles/wave  {100 * x / tot:5.1f} %")
