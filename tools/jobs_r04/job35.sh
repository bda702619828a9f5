#!/bin/bash
# round 4: why 720p ran at 14.1 ms in job34 and at 9.7 in job32: the library, the flags or the box
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']), {k: v['ms'] for k, v in j['kernels'].items() if v['ms'] > 0.05})"; }
: > $O/p720.txt
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3 --width 1280 --height 720 2>/dev/null | pr "product no-cpu" | tee -a $O/p720.txt
timeout -k 10 300 python bench.py --no-stress --no-e2e --no-sweep --steps 10 --verify-frames 64 --cpu-seconds 1 --width 1280 --height 720 2>/dev/null | pr "product with cpu legs" | tee -a $O/p720.txt
MI_RTJ_LIB=$L timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3 --width 1280 --height 720 2>/dev/null | pr "experiments build no-cpu" | tee -a $O/p720.txt
MI_RTJ_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3 --width 1280 --height 720 2>/dev/null | pr "product classic no-cpu" | tee -a $O/p720.txt
done
