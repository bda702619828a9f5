#!/usr/bin/env python3
"""Writes tests/golden/dv_golden.json: digests of DIF frames made by oracle/dv_oracle.c's encoder from its synthetic
pictures, and of the pictures its decoder makes of them and of a frame of random bytes.  (The DV statement has no
reference to be generated from — PARITY UNPINNED, oracle/dv_oracle.h; the fixture only keeps the statement from
drifting unnoticed.)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import dvlib as D  # noqa: E402

out = {"frames": [], "fuzz_seed": 20261005}
for n, seed, amp, flags in [(0, 1, 0, 0), (3, 7, 4, 3), (5, 11, 12, 3), (9, 2, 40, 1)]:
    dif = D.encode(D.synth(n, seed, amp), flags)
    out["frames"].append({"n": n, "seed": seed, "amp": amp, "flags": flags, "dif_sha256": hashlib.sha256(dif.tobytes()).hexdigest(),
                          "pic_sha256": hashlib.sha256(D.decode(dif).tobytes()).hexdigest()})
rng = np.random.default_rng(out["fuzz_seed"])
out["fuzz_pic_sha256"] = hashlib.sha256(D.decode(rng.integers(0, 256, D.FRAME_BYTES, dtype=np.uint8)).tobytes()).hexdigest()
json.dump(out, open(os.path.join(HERE, "dv_golden.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
