#!/usr/bin/env python3
"""Throughput of the colour stage (N2) on device-resident frames: GB/s of algorithmic traffic
(1.5 B read + bpp written per pixel; the 32-bit formats also re-read the 4 B they merge into)."""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("gmerlin-avdecoder_amd")

w, h, n, reps = 1920, 1088, 128, 20
dev = P.MiRtj()
fsz = w * h * 3 // 2
d_in = dev.synth(w, h, 0, n)
res = {}
for fmt, name, bpp in [(0, "rgb32", 4), (1, "bgr32", 4), (2, "rgb24", 3), (3, "bgr24", 3), (4, "rgb16", 2)]:
    pitch = w * bpp
    d_out = dev.alloc(pitch * h * n)
    dev.memset(d_out, 0, pitch * h * n)
    for _ in range(3):
        dev.to_rgb(fmt, w, h, n, d_in, fsz, d_out, pitch, pitch * h)
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.to_rgb(fmt, w, h, n, d_in, fsz, d_out, pitch, pitch * h)
    dev.sync()
    dt = (time.perf_counter() - t0) / reps
    alg = n * w * h * (1.5 + bpp + (4 if bpp == 4 else 0))
    res[name] = {"ms": round(dt * 1e3, 4), "gbs": round(alg / dt / 1e9, 1), "frac_of_8TBs": round(alg / dt / 8e12, 3),
                 "mpix_s": round(n * w * h / dt / 1e6)}
    dev.free(d_out)
print(json.dumps(res))
