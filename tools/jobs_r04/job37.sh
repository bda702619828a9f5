#!/bin/bash
# round 4: index of launch k + 1 next to the transform of launch k (MI_RTJ_OVERLAP=1) at 16,384 per launch, by content
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), 'ms/step', j['ms_per_step'], {k: v['ms'] for k, v in j['kernels'].items() if v['ms'] > 0.05})"; }
B="--no-cpu --no-e2e --no-sweep --no-stress --content hash"
: > $O/overlap_16384.txt
for amp in 8 32 64; do for ov in 0 1; do
MI_RTJ_OVERLAP=$ov timeout -k 10 300 python bench.py $B --amp $amp --steps 10 --warmup 6 2>/dev/null | pr "amp$amp overlap=$ov" | tee -a $O/overlap_16384.txt
done; done
