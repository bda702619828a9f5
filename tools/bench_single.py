#!/usr/bin/env python3
"""End-to-end rate of the one-packet path (mi_rtj_decode: host packet in, host planes out, PCIe both
ways, synchronous) — what the bgav_video_decoder_t wrapper uses.  Not the bench value; DESIGN.md §5."""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = importlib.import_module("gmerlin-avdecoder_amd")
res = {}
for (w, h) in ((320, 240), (1920, 1088), (3840, 2160)):
    dev = P.MiRtj()
    n = 32
    d_fr = dev.synth(w, h, 0, n)
    d_st, po, pl = dev.encode(w, h, 255, n, d_fr)
    dev.sync()
    pkts = [dev.d2h(d_st, int(pl[i]), offset=int(po[i])) for i in range(n)]
    out = np.zeros(w * h * 3 // 2, np.uint8)
    for p in pkts[:4]:
        dev.decode(p, out)
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        for p in pkts:
            dev.decode(p, out)
    dt = (time.perf_counter() - t0) / (reps * n)
    t0 = time.perf_counter()
    for _ in range(reps):
        for p in pkts:
            dev.decode_nocopy(p)
    dt2 = (time.perf_counter() - t0) / (reps * n)
    res[f"{w}x{h}"] = {"ms_per_frame": round(dt * 1e3, 4), "fps": round(1 / dt, 1),
                       "nocopy_ms_per_frame": round(dt2 * 1e3, 4), "nocopy_fps": round(1 / dt2, 1),
                       "host_bytes_per_frame": int(np.mean([p.size for p in pkts]) + out.size)}
    dev.close()
print(json.dumps(res))
