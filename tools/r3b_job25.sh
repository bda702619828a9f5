#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" python - "$label" <<'PY' | tee -a $O/e2e_two_streams2.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(1920, 1088, packets=64, repeat=32, depth=int(os.environ.get("DEPTH", "12")), flavours=("_pipe",), two_streams=True)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:44s} one stream {one.get('fps')}   two streams on two threads {r.get('two_streams_two_threads',{}).get('fps')}")
PY
}
run "default" X=1
run "one copy-out stream per session" MI_RTJ_OUT_STREAMS=1
run "no copy out (skip=1)" MI_RTJ_EXP_SKIP=1
run "copy out only (skip=6)" MI_RTJ_EXP_SKIP=6
run "no kernels (skip=2)" MI_RTJ_EXP_SKIP=2
run "pairs depth 6" DEPTH=6
run "GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
run "HSA_ENABLE_SDMA=0 (blit kernels)" HSA_ENABLE_SDMA=0
