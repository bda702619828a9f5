#!/bin/bash
# round 4: DV decoder debug build: per-lane coefficients and state against the checker's
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
MI_DV_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_dv_dbg.so timeout -k 5 120 python - > $O/dv_dbg.log 2>&1 <<'PY'
import sys, importlib, ctypes as C, numpy as np
sys.path.insert(0, "tests")
import dvlib as D
dv = importlib.import_module("gmerlin-avdecoder_amd.dv")
d = dv.MiDv(0)
L = d.L
L.mi_dv_debug_buffer.argtypes = [C.c_void_p]
D.lib().dvo_segment_coefs.argtypes = [D.u8p, C.c_int, C.c_int, C.c_void_p]
for amp, flags in [(0, 0), (8, 3)]:
    f = D.encode(D.synth(1, 5, amp), flags)
    dbg = d.alloc(135 * 64 * 72 * 2)
    L.mi_dv_debug_buffer(dbg)
    got = d.decode_frames(f)[0]
    st = d.d2h(dbg, 135 * 64 * 72 * 2).view(np.int16).reshape(135, 64, 72)
    nbad = 0
    for S in range(270):
        seq, slot = divmod(S, 27)
        want = np.zeros((30, 64), np.int16)
        D.lib().dvo_segment_coefs(D.p8(f), seq, slot, want.ctypes.data)
        want[:, 0] += 4  # the kernel carries DESCALE's rounding term on the DC
        g = st[S // 2, 30 * (S % 2):30 * (S % 2) + 30]
        for b in range(30):
            if not np.array_equal(g[b, :64], want[b]):
                nbad += 1
                if nbad <= 6:
                    k = np.flatnonzero(g[b, :64] != want[b])
                    print(f"amp {amp} seg {S} blk {b} (mb {b//6} j {b%6}): {k.size} coefs differ at nat {k[:10]} got {g[b, k[:6]]} want {want[b, k[:6]]} "
                          f"state pos {g[b,64]} p {g[b,65]} fin {g[b,66]} npart {g[b,67]} mode {g[b,68]} cls {g[b,69]} qno {g[b,70]} mlen {g[b,71]}", flush=True)
    print("amp", amp, "blocks with wrong coefficients:", nbad, "of 8100", flush=True)
    w2 = D.decode(f)
    print("pixels differ", int((got != w2).sum()), flush=True)
    bad = np.flatnonzero(got != w2)
    yb = bad[bad < 720 * 480]
    print("luma: by column mod 8", np.bincount(yb % 720 % 8, minlength=8), "by row mod 8", np.bincount(yb // 720 % 8, minlength=8), flush=True)
    print("luma: by 32-pixel column", np.bincount(yb % 720 // 32, minlength=23), flush=True)
    cb = bad[bad >= 720 * 480] - 720 * 480
    print("chroma: count", cb.size, "by column mod 8", np.bincount(cb % 180 % 8, minlength=8), "plane", np.bincount(cb // (180 * 480), minlength=2), flush=True)
    dd = got.astype(int) - w2.astype(int)
    print("difference histogram", {int(v): int(c) for v, c in zip(*np.unique(dd[bad], return_counts=True))}, flush=True)
    i0 = int(yb[0]); r0, c0 = divmod(i0, 720)
    print("first bad luma pixel", r0, c0, "got block rows:", flush=True)
    br, bc = r0 // 8 * 8, c0 // 8 * 8
    print(got[:720*480].reshape(480, 720)[br:br+8, bc:bc+8]); print(w2[:720*480].reshape(480, 720)[br:br+8, bc:bc+8], flush=True)
    d.free(dbg)
d.close()
PY
echo "rc=$?"; cat $O/dv_dbg.log
