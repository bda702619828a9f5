#!/bin/bash
# End-to-end A/B on the GPU box: one stream through the default build of the wrapper (tests/harness/plugin_harness.c in
# bench mode, steady state: the first lap of the packet list is not timed), by session switches.
#   bash tools/e2e_ab.sh > gpurun_out/<tag>/e2e_ab.txt
# Columns: pictures per second, microseconds per picture, where the submitting thread's time went (MI_RTJ_PIPE_STATS=1).
# (profiles/r03/e2e_index_and_copy_groups.txt, e2e_steady_state.txt were made this way.)
cd "$(dirname "$0")/.."
run() { # label, env...
  local label=$1; shift
  env "$@" MI_RTJ_PIPE_STATS=1 python - "$label" <<'PY'
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "12")), flavours=("_pipe",), two_streams=bool(os.environ.get("TWO")))
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
two = r.get('two_streams_two_threads', {}).get('fps', '')
print(f"{sys.argv[1]:56s} {one.get('fps')}  us/picture {round(1e6/one['fps'],1)}  {one.get('pipe_stats',{}).get('us_per_picture')} {two}")
PY
}
K4="W=3840 H=2160 PK=24 REP=16"
for rep in 1 2; do
run "1080p default (12 in flight, fours)" X=1
run "1080p idx 1, out 2, depth 6 (round 3's first session)" MI_RTJ_IDX_GROUP=1 MI_RTJ_OUT_GROUP=2 DEPTH=6
run "1080p idx 2, out 2, depth 6" DEPTH=6
run "1080p idx 4, out 4, depth 8 (two groups in flight)" MI_RTJ_IDX_GROUP=4 MI_RTJ_OUT_GROUP=4 DEPTH=8
run "1080p idx 2, out 4, depth 12" MI_RTJ_IDX_GROUP=2
done
EXPL=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so  # python gmerlin-avdecoder_amd/build.py --experiments (the switches below only exist in that build; the harness loads libmi_rtjpeg.so by rpath: LD_PRELOAD the other)
run "1080p no copy out (MI_RTJ_EXP_SKIP=1)" LD_PRELOAD=$EXPL MI_RTJ_EXP_SKIP=1
run "1080p copy out only (MI_RTJ_EXP_SKIP=6)" LD_PRELOAD=$EXPL MI_RTJ_EXP_SKIP=6
run "4K default" $K4
run "4K pairs, depth 6" $K4 DEPTH=6
run "320x240 default" W=320 H=240 REP=200
run "1080p two streams on two threads" TWO=1
