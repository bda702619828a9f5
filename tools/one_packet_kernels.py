#!/usr/bin/env python3
"""Per-kernel device time of ONE packet per launch (the path of the plugin's sessions): python tools/one_packet_kernels.py [w h]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("gmerlin-avdecoder_amd")
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1088)
dev = P.MiRtj(0)
n = 32
d_fr = dev.synth(w, h, 0, n, seed=12345, amp=8)
d_st, po, pl = dev.encode(w, h, 255, n, d_fr)
dev.sync()
pkts = [dev.d2h(d_st, int(pl[i]), offset=int(po[i])) for i in range(n)]
pipe = dev.pipe(depth=6, coded_w=w, coded_h=h)
def lap():
    got = nxt = 0
    while got < n:
        while nxt < n and pipe.room() > 0:
            pipe.submit(pkts[nxt], nxt); nxt += 1
        pipe.next(); got += 1
lap()
pipe.profile(True)
lap(); lap()
ms, launches = pipe.times()
print(json.dumps({"geometry": f"{w}x{h}", "packets": launches, "us_per_packet": {k: round(v / launches * 1e3, 2) for k, v in ms.items() if v > 0},
                  "sum_us": round(sum(ms.values()) / launches * 1e3, 2)}))
