#!/bin/bash
# round 4: small launches with shorter luma waves (product builds with -DMIRTJ_SPLIT_LUMA_SG=1 / 2; 3 is the default)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), {k: v['ms'] for k, v in j['kernels'].items() if v['ms'] > 0.02})"; }
: > $O/small_launch_sg.txt
for n in 256 512 1024 2048; do for sg in 3 2 1; do
  if [ $sg -eq 3 ]; then unset MI_RTJ_LIB; else export MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_sg$sg.so; fi
  timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --frames $n --steps 64 --warmup 8 2>/dev/null | pr "frames=$n super groups per luma wave=$sg" | tee -a $O/small_launch_sg.txt
done; done
