#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_plugin_harness.py tests/test_bench_launcher.py -m gpu -x -q > $O/t_groups.log 2>&1; echo "pytest rc=$?"; tail -5 $O/t_groups.log
grep -q " passed" $O/t_groups.log || exit 1
run() { # label, env...
  local label=$1; shift
  env "$@" MI_RTJ_PIPE_STATS=1 python - "$label" <<'PY' | tee -a $O/e2e_idx_groups.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "6")), flavours=("_pipe",), two_streams=False)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:56s} {one.get('fps')}  us/picture {round(1e6/one['fps'],1)}  {one.get('pipe_stats',{}).get('us_per_picture')}")
PY
}
K4="W=3840 H=2160 PK=24 REP=16"
for rep in 1 2; do
run "1080p idx 1, out 2, depth 6 (round 3 so far)" MI_RTJ_IDX_GROUP=1
run "1080p idx 2, out 2, depth 6 (default)" X=1
run "1080p idx 2, out 2, depth 8" DEPTH=8
run "1080p idx 4, out 2, depth 8" MI_RTJ_IDX_GROUP=4 DEPTH=8
run "1080p idx 4, out 4, depth 8" MI_RTJ_IDX_GROUP=4 MI_RTJ_OUT_GROUP=4 DEPTH=8
run "1080p idx 4, out 4, depth 12" MI_RTJ_IDX_GROUP=4 MI_RTJ_OUT_GROUP=4 DEPTH=12
run "1080p idx 2, out 4, depth 12" MI_RTJ_IDX_GROUP=2 MI_RTJ_OUT_GROUP=4 DEPTH=12
done
run "1080p idx 2 skip=1 (no copy out)" MI_RTJ_EXP_SKIP=1
run "1080p idx 4 skip=1 depth 8" MI_RTJ_EXP_SKIP=1 MI_RTJ_IDX_GROUP=4 DEPTH=8
run "1080p idx 1 skip=1" MI_RTJ_EXP_SKIP=1 MI_RTJ_IDX_GROUP=1
run "4K idx 1" $K4 MI_RTJ_IDX_GROUP=1
run "4K idx 2 (default)" $K4
run "4K idx 2 skip=1" $K4 MI_RTJ_EXP_SKIP=1
run "4K idx 1 skip=1" $K4 MI_RTJ_EXP_SKIP=1 MI_RTJ_IDX_GROUP=1
run "320x240 idx 1" W=320 H=240 PK=64 REP=200 MI_RTJ_IDX_GROUP=1
run "320x240 idx 2" W=320 H=240 PK=64 REP=200
run "320x240 idx 4 depth 8" W=320 H=240 PK=64 REP=200 MI_RTJ_IDX_GROUP=4 DEPTH=8
