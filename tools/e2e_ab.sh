#!/bin/bash
# End-to-end A/B on the GPU box (one 1080p stream through the default build of the wrapper).
#   bash tools/e2e_ab.sh > gpurun_out/<tag>/e2e_ab.txt
cd "$(dirname "$0")/.."
run() { # label, env...
  local label=$1; shift
  env "$@" python - "$label" <<'PY'
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "6")), flavours=("_pipe",), two_streams=bool(os.environ.get("TWO")))
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:66s} {one.get('fps')}", r.get('two_streams_two_threads', {}).get('fps', ''))
PY
}
for rep in 1 2 3; do
run "pairs, depth 6" X=1
run "pairs, depth 8" DEPTH=8
run "fours, depth 8" MI_RTJ_OUT_GROUP=4 DEPTH=8
run "fours, depth 12" MI_RTJ_OUT_GROUP=4 DEPTH=12
run "fours, depth 16" MI_RTJ_OUT_GROUP=4 DEPTH=16
done
run "4K: pairs, depth 6" W=3840 H=2160 PK=24 REP=8 DEPTH=6
run "4K: fours, depth 8" W=3840 H=2160 PK=24 REP=8 DEPTH=8 MI_RTJ_OUT_GROUP=4
run "4K: fours, depth 12" W=3840 H=2160 PK=24 REP=8 DEPTH=12 MI_RTJ_OUT_GROUP=4
