#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_plugin_harness.py -m gpu -x -q > $O/t_groups.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_groups.log
run() { # label, env...
  local label=$1; shift
  env "$@" MI_RTJ_PIPE_STATS=1 python - "$label" <<'PY' | tee -a $O/e2e_idx_groups2.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "6")), flavours=("_pipe",), two_streams=False)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:56s} {one.get('fps')}  us/picture {round(1e6/one['fps'],1)}  {one.get('pipe_stats',{}).get('us_per_picture')}")
PY
}
K4="W=3840 H=2160 PK=24 REP=16"
run "4K idx 2 out 2 depth 6" $K4
run "4K idx 2 out 4 depth 12" $K4 MI_RTJ_OUT_GROUP=4 DEPTH=12
run "4K idx 4 out 4 depth 12" $K4 MI_RTJ_OUT_GROUP=4 MI_RTJ_IDX_GROUP=4 DEPTH=12
run "4K idx 2 out 2 depth 12" $K4 DEPTH=12
run "1080p idx 2 out 4 depth 12" MI_RTJ_OUT_GROUP=4 DEPTH=12
run "1080p idx 2 out 4 depth 16" MI_RTJ_OUT_GROUP=4 DEPTH=16
run "1080p idx 2 out 2 depth 12" DEPTH=12
run "1080p idx 2 out 2 depth 6" DEPTH=6
