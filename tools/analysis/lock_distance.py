"""How far a walk that starts at an arbitrary byte (as if a macroblock began there) runs before it falls into step
with the true block chain, on the synthetic content at a given noise amplitude: the lead a speculative walker needs.

    python tools/analysis/lock_distance.py [amp] [Q] [starts]      (CPU only; test infrastructure)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import rtjlib as R

amp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 255
nstart = int(sys.argv[3]) if len(sys.argv) > 3 else 300
w, h = 1920, 1088
_, _, lb8, cb8, _, _ = R.oracle_tables(Q)
pkt = R.OracleEncoder(w, h, Q).encode(R.synth_frame(w, h, 0, amp=amp))
offs = R.OracleDecoder().block_offsets(pkt).astype(np.int64) - 12
body = pkt[12:].astype(np.int64)
n = body.size
sb = np.where(body > 127, body - 256, body)
true_mb = set(int(x) for x in offs[0:-1:6])
true_blk = set(int(x) for x in offs[:-1])
by_block = lb8 == cb8


def block_len(p, bt8):
    if p >= n:
        return 64
    if body[p] == 255:
        return 1
    q = p + 1 + bt8
    slots = 63 - bt8
    while slots > 0:
        v = sb[q] if q < n else 0
        slots -= (v - 63) if v > 63 else 1
        q += 1
    return q - p


rng = np.random.default_rng(1)
dist = []
for s in rng.integers(0, n - 40000, nstart):
    p, ph = int(s), 0
    locked = None
    while p - s < 32768:
        if (by_block and p in true_blk) or (not by_block and ph == 0 and p in true_mb):
            locked = p - s
            break
        p += block_len(p, cb8 if ph >= 4 else lb8)
        ph = (ph + 1) % 6
    dist.append(locked if locked is not None else 1 << 30)
d = np.array(dist)
print(f"amp {amp} Q {Q}: packet {n} bytes, {n / (offs.size - 1):.1f} bytes per block, lb8 {lb8} cb8 {cb8}")
for q in (50, 75, 90, 95, 99, 99.7):
    print(f"  {q:5.1f} % of the walks are in step within {int(np.percentile(d, q))} bytes")
for lead in (768, 1536, 3072, 4096, 6144, 8192, 16384):
    print(f"  lead {lead:5d}: {(d <= lead).mean() * 100:6.2f} % in step")
