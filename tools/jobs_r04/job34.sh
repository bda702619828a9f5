#!/bin/bash
# round 4: luma waves per XCD and frame by shape (the split form; MI_RTJ_LUMA_WAVES exists in the experiments build)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3"
true
for lw in 0 7 9 11 15 21; do
MI_RTJ_LIB=$L MI_RTJ_LUMA_WAVES=$lw timeout -k 10 300 python bench.py $B --width 3840 --height 2160 --frames 4096 2>/dev/null | pr "4K lw=$lw (0: by the rule, 15; 43 super groups per XCD, cw=4)" | tee -a $O/lw_by_shape.txt
done
for lw in 0 1 3 5; do
MI_RTJ_LIB=$L MI_RTJ_LUMA_WAVES=$lw timeout -k 10 300 python bench.py $B --width 1280 --height 720 2>/dev/null | pr "720p lw=$lw (0: by the rule, 2; 5 super groups per XCD, cw=1)" | tee -a $O/lw_by_shape.txt
done
for lw in 0 1 3; do
MI_RTJ_LIB=$L MI_RTJ_LUMA_WAVES=$lw timeout -k 10 300 python bench.py $B --width 720 --height 576 --frames 32768 2>/dev/null | pr "720x576 lw=$lw (0: by the rule, 2; 3 super groups per XCD, cw=1)" | tee -a $O/lw_by_shape.txt
done
