#!/usr/bin/env python3
"""What the host link of this box sustains for one picture-sized pinned copy at a time, device to host and host to
device (the end-to-end session is bounded by the copy out: one 1080p picture is 3.13 MB)."""
import json, time, torch
out = {}
for name, nbytes in (("1080p picture (3.13 MB)", 1920 * 1088 * 3 // 2), ("4K picture (12.4 MB)", 3840 * 2160 * 3 // 2), ("1080p packet (0.59 MB)", 588307)):
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    s = torch.cuda.Stream()
    for direction in ("d2h", "h2d"):
        with torch.cuda.stream(s):
            for _ in range(20):
                (h.copy_(d, non_blocking=True) if direction == "d2h" else d.copy_(h, non_blocking=True))
            s.synchronize()
            t0 = time.perf_counter()
            n = 400
            for _ in range(n):
                (h.copy_(d, non_blocking=True) if direction == "d2h" else d.copy_(h, non_blocking=True))
            s.synchronize()
            dt = time.perf_counter() - t0
        out[f"{name} {direction}"] = {"us_per_copy": round(dt / n * 1e6, 1), "GB_per_s": round(nbytes * n / dt / 1e9, 1)}
print(json.dumps(out))
