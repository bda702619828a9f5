#!/bin/bash
# what bounds the copy out of a session: copy in (skip bit 4) and kernels (bit 2) switched off, pairs / fours
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" python - "$label" <<'PY' | tee -a $O/e2e_copy_bounds.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "6")), flavours=("_pipe",), two_streams=False)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:60s} {one.get('fps')}")
PY
}
for sk in 0 2 6 1; do
run "1080p skip=$sk pairs depth 6" MI_RTJ_EXP_SKIP=$sk
run "1080p skip=$sk fours depth 8" MI_RTJ_EXP_SKIP=$sk MI_RTJ_OUT_GROUP=4 DEPTH=8
run "1080p skip=$sk singles depth 6" MI_RTJ_EXP_SKIP=$sk MI_RTJ_OUT_GROUP=1
done
for sk in 0 2 6; do
run "4K skip=$sk pairs depth 6" W=3840 H=2160 PK=24 REP=8 MI_RTJ_EXP_SKIP=$sk
run "4K skip=$sk singles depth 6" W=3840 H=2160 PK=24 REP=8 MI_RTJ_EXP_SKIP=$sk MI_RTJ_OUT_GROUP=1
run "4K skip=$sk fours depth 8" W=3840 H=2160 PK=24 REP=8 MI_RTJ_EXP_SKIP=$sk MI_RTJ_OUT_GROUP=4 DEPTH=8
done
python tools/pcie_probe.py | tee $O/pcie_probe.json
