#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_plugin_harness.py tests/test_gpu_sessions.py -m gpu -x -q > $O/t_sess2.log 2>&1; echo "pytest rc=$?"; tail -3 $O/t_sess2.log
run() { # label, env...
  local label=$1; shift
  env "$@" MI_RTJ_PIPE_STATS=1 python - "$label" <<'PY' | tee -a $O/e2e_after_event_change.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "12")), flavours=("_pipe",), two_streams=False)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:56s} {one.get('fps')}  us/picture {round(1e6/one['fps'],1)}  {one.get('pipe_stats',{}).get('us_per_picture')}")
PY
}
for rep in 1 2 3; do run "1080p default (depth 12, fours)" X=1; done
run "1080p skip=1" MI_RTJ_EXP_SKIP=1
run "4K default" W=3840 H=2160 PK=24 REP=16
run "320x240 default" W=320 H=240 REP=200
