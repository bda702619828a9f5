"""What the blocks of the bench content look like, wave by wave, as k_decode groups them (64 blocks per round:
32 macroblocks x (upper luma | lower luma | chroma)).  Answers, from the stream itself: how often could a whole wave
take a cheaper transform variant (columns 6-7 empty, rows 6-7 empty, low 4x4, ...), how long the parse runs.

    python tools/analysis/block_shapes.py [w h Q amp frames]

CPU only (oracle encoder + s2b); test infrastructure, not product."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import ctypes as C
import rtjlib as R

w, h, Q, amp, nfr = (int(x) for x in (sys.argv[1:6] + [1920, 1088, 255, 8, 2][len(sys.argv) - 1:]))
liqt, ciqt, lb8, cb8, _, _ = R.oracle_tables(Q)
L = R.oracle()
dec = R.OracleDecoder()
ZZ = [0, 8, 1, 2, 9, 16, 24, 17, 10, 3, 4, 11, 18, 25, 32, 40, 33, 26, 19, 12, 5, 6, 13, 20, 27, 34, 41, 48, 56, 49, 42, 35,
      28, 21, 14, 7, 15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62,
      55, 63]
stats = {}
def add(k, v=1):
    stats[k] = stats.get(k, 0) + v
for n in range(nfr):
    pkt = R.OracleEncoder(w, h, Q).encode(R.synth_frame(w, h, n, amp=amp))
    offs = dec.block_offsets(pkt)
    body = pkt  # block_offsets counts from the packet start (header included)
    nmb = (w // 16) * (h // 16)
    coef = np.zeros((nmb * 6, 64), np.int16)
    blen = np.zeros(nmb * 6, np.int32)
    lq = (C.c_int32 * 64)(*liqt.tolist()); cq = (C.c_int32 * 64)(*ciqt.tolist())
    blk = (C.c_int16 * 64)()
    pad = np.concatenate([body, np.zeros(128, np.uint8)])
    for b in range(nmb * 6):
        o = int(offs[b]); blen[b] = int(offs[b + 1]) - o
        ch = (b % 6) >= 4
        if pad[o] == 255:
            continue
        L.rtjo_s2b(R._ptr(pad[o:o + 70].copy()), 70, cb8 if ch else lb8, cq if ch else lq, blk)
        coef[b] = np.frombuffer(blk, dtype=np.int16)
    nz = (coef != 0).reshape(-1, 8, 8)  # [block][row][col]
    ngroups = (nmb + 31) // 32
    for g in range(ngroups):
        mbs = range(g * 32, min(nmb, g * 32 + 32))
        for part in range(3):
            if part < 2:
                ids = [6 * m + 2 * part + k for m in mbs for k in (0, 1)]
            else:
                ids = [6 * m + 4 for m in mbs] + [6 * m + 5 for m in mbs]
            z = nz[ids].any(axis=0)  # [row][col] over the wave
            kind = "luma" if part < 2 else "chroma"
            add(kind + " waves")
            add(kind + " maxlen", int(blen[ids].max()))
            add(kind + " sumlen", int(blen[ids].sum())); add(kind + " blocks", len(ids))
            for name, cond in (("cols 6-7 empty", not z[:, 6:].any()), ("rows 6-7 empty", not z[6:, :].any()),
                               ("cols 6-7 and rows 6-7 empty", not z[:, 6:].any() and not z[6:, :].any()),
                               ("cols 4-7 empty", not z[:, 4:].any()), ("rows 4-7 empty", not z[4:, :].any()),
                               ("low 4x4", not z[:, 4:].any() and not z[4:, :].any()),
                               ("low 3x3", not z[:, 3:].any() and not z[3:, :].any()),
                               ("low 2x2", not z[:, 2:].any() and not z[2:, :].any()),
                               ("rows 5-7 empty", not z[5:, :].any()), ("cols 5-7 empty", not z[:, 5:].any()),
                               ("row 7 empty", not z[7:, :].any()), ("col 7 empty", not z[:, 7:].any())):
                if cond:
                    add(kind + " " + name)
            per_blk = nz[ids]
            add(kind + " blocks with cols 6-7 empty", int((~per_blk[:, :, 6:].any(axis=(1, 2))).sum()))
            add(kind + " blocks DC only", int((~per_blk.reshape(len(ids), 64)[:, 1:].any(axis=1)).sum()))
print(f"{w}x{h} Q={Q} amp={amp} frames={nfr} lb8={lb8} cb8={cb8}")
for kind in ("luma", "chroma"):
    nw = stats[kind + " waves"]
    print(f"{kind}: {nw} waves, avg block {stats[kind+' sumlen']/stats[kind+' blocks']:.2f} bytes, avg longest of a wave {stats[kind+' maxlen']/nw:.1f}")
    for k in sorted(stats):
        if k.startswith(kind + " ") and ("empty" in k or "low" in k or "DC" in k):
            denom = stats[kind + " blocks"] if "blocks" in k else nw
            print(f"   {k[len(kind)+1:]:40s} {stats[k]/denom*100:6.1f} %")
