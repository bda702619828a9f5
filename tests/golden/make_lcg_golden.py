#!/usr/bin/env python3
"""Writes tests/golden/lcg_golden.json: frames of SURVEY.md section 8d's generator (tests/rtjlib.py synth_frame_lcg: the
LCG noise of BASELINE.md section 2's CPU probe) encoded and decoded by the REFERENCE's own lib/RTjpeg.c (oracle/_ref,
built from /root/reference by oracle/Makefile — run this where that exists): packet sizes and digests of packets and
planes.  The GPU generator, encoder and decoder are checked against them (tests/test_gpu_lcg.py)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import rtjlib as R  # noqa: E402

assert R.have_reference(), "needs oracle/_ref/librtjpeg_ref.so (make -C oracle, with /root/reference present)"
out = {"generator": "synth_frame_lcg, seed 12345", "frames": []}
for w, h, Q, amp, frames in [(320, 240, 255, 8, (0, 1, 7)), (1920, 1088, 255, 8, (0, 5)), (1920, 1088, 255, 64, (2,)), (640, 368, 128, 8, (3,))]:
    for n in frames:
        enc = R.RefCodec()
        enc.setup_encoder(w, h, Q)
        f = R.synth_frame_lcg(w, h, n, seed=12345, amp=amp)
        pkt = enc.encode(f)
        pic = np.zeros(w * h * 3 // 2, np.uint8)
        R.RefCodec().decode(pkt, pic)
        out["frames"].append({"w": w, "h": h, "Q": Q, "amp": amp, "n": n, "frame": R.digest(f), "packet_bytes": int(pkt.size),
                              "packet": R.digest(pkt), "planes": R.digest(pic)})
json.dump(out, open(os.path.join(HERE, "lcg_golden.json"), "w"), indent=1)
print(json.dumps(out["frames"], indent=1))
