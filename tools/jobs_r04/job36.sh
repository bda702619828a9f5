#!/bin/bash
# round 4: super groups per luma wave (product builds with -DMIRTJ_SPLIT_LUMA_SG=n; 3 is the default) by picture size
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3"
: > $O/sg_by_shape.txt
for shape in "--width 3840 --height 2160 --frames 4096" "--width 1280 --height 720" "--width 720 --height 576 --frames 32768" "--width 1920 --height 1088"; do
for sg in 2 3 4 6; do
  if [ $sg -eq 3 ]; then unset MI_RTJ_LIB; else export MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_sg$sg.so; fi
  timeout -k 10 300 python bench.py $B $shape 2>/dev/null | pr "$shape: $sg super groups per luma wave" | tee -a $O/sg_by_shape.txt
done; done
