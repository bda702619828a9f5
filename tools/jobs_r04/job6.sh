#!/bin/bash
# round 4: DV decoder first light, under short timeouts, everything logged
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 5 90 python - > $O/dv_first.log 2>&1 <<'PY'
import sys, importlib, numpy as np, time
sys.path.insert(0, "tests")
import dvlib as D
dv = importlib.import_module("gmerlin-avdecoder_amd.dv")
d = dv.MiDv(0)
print("created", flush=True)
for amp, flags in [(0, 0), (8, 3), (40, 3)]:
    f = D.encode(D.synth(1, 5, amp), flags)
    t = time.time()
    got = d.decode_frames(f)[0]
    print("decoded", amp, flags, round(time.time() - t, 3), flush=True)
    want = D.decode(f)
    bad = np.flatnonzero(got != want)
    print("amp", amp, "flags", flags, "differ", bad.size, "first", bad[:8], flush=True)
    if bad.size:
        y = bad[bad < 720 * 480]
        print("  luma bad", y.size, "rows", np.unique(y // 720)[:10], "cols", np.unique(y % 720)[:10], flush=True)
rng = np.random.default_rng(3)
f = rng.integers(0, 256, D.FRAME_BYTES, dtype=np.uint8)
got = d.decode_frames(f)[0]; want = D.decode(f)
print("fuzz differ", int((got != want).sum()), flush=True)
d.close()
PY
echo "rc=$?"; cat $O/dv_first.log
