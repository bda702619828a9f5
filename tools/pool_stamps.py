#!/usr/bin/env python3
"""Timing builds only (python gmerlin-avdecoder_amd/build.py --experiments; MI_RTJ_LIB=.../libmi_rtjpeg_exp.so): where a
pooling chroma wave of k_decode_split spends its time — shader-clock ticks per section, from s_memtime stamps summed
over all waves (csrc/rtj_decode_chroma.h, MIRTJ_PSTAMP)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("gmerlin-avdecoder_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = P.MiRtj(0)
w, h, Q = 1920, 1088, 255
d_fr = dev.synth(w, h, 0, n, seed=12345, amp=8)
d_st, po, pl = dev.encode(w, h, Q, n, d_fr)
dev.sync()
dev.free(d_fr)
fsz = w * h * 3 // 2
hdr0 = dev.d2h(d_st, 12, offset=int(po[0]))
d_out = dev.alloc(fsz * n)
plan = dev.plan(np.tile(hdr0, (n, 1)), po, pl, np.arange(n, dtype=np.uint64) * np.uint64(fsz))
for _ in range(2):
    plan.decode(d_st, d_out)
dev.sync()
L = P.load()
out = (C.c_ulonglong * 16)()
L.mi_rtj_debug_pool_stamps(out)
plan.decode(d_st, d_out)
dev.sync()
L.mi_rtj_debug_pool_stamps(out)
names = ["prologue", "bytes + next addresses + loads issued", "classification", "hand-over + parse", "transform + rows to LDS", "row stores", "counted wait + copies"]
waves = out[7]
tot = sum(out[i] for i in range(7))
print(f"{n} pictures, {waves} pooling waves, {tot / max(waves, 1):.0f} ticks per wave (s_memtime, 100 MHz reference clock)")
for i, nm in enumerate(names):
    print(f"  {nm:42s} {out[i] / max(waves, 1):10.1f} ticks per wave  {100.0 * out[i] / max(tot, 1):5.1f} %")
