#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" python - "$label" <<'PY' | tee -a $O/e2e_two_streams.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "12")), flavours=("_pipe",), two_streams=True)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:40s} one stream {one.get('fps')}   two streams on two threads {r.get('two_streams_two_threads',{}).get('fps')}")
PY
}
run "1080p sync" X=1
run "1080p query+yield" MI_RTJ_WAIT=query
run "1080p sync" X=1
run "1080p query+yield" MI_RTJ_WAIT=query
run "320x240 sync" W=320 H=240 REP=200
run "320x240 query+yield" W=320 H=240 REP=200 MI_RTJ_WAIT=query
