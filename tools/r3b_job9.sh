#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" MI_RTJ_PIPE_STATS=1 python - "$label" <<'PY' | tee -a $O/e2e_steady.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "6")), flavours=("_pipe",), two_streams=False)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:50s} {one.get('fps')}  us/picture {round(1e6/one['fps'],1)}  {one.get('pipe_stats',{}).get('us_per_picture')}")
PY
}
K4="W=3840 H=2160 PK=24 REP=16"
run "1080p pairs depth 6" X=1
run "1080p pairs depth 8" DEPTH=8
run "1080p singles depth 6" MI_RTJ_OUT_GROUP=1
run "1080p fours depth 8" MI_RTJ_OUT_GROUP=4 DEPTH=8
run "1080p fours depth 12" MI_RTJ_OUT_GROUP=4 DEPTH=12
run "1080p pairs skip=1 (no copy out)" MI_RTJ_EXP_SKIP=1
run "1080p pairs skip=6 (copy out only)" MI_RTJ_EXP_SKIP=6
run "4K pairs depth 6" $K4
run "4K pairs depth 8" $K4 DEPTH=8
run "4K singles depth 6" $K4 MI_RTJ_OUT_GROUP=1
run "4K pairs skip=1" $K4 MI_RTJ_EXP_SKIP=1
run "4K pairs skip=6" $K4 MI_RTJ_EXP_SKIP=6
for d in 4 6 8; do MI_RTJ_DEPTH_OVERRIDE=$d python bench.py --config streams4k --no-cpu --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams4k depth $d', d['value'])" | tee -a $O/e2e_steady.txt; done
