#!/bin/bash
# round 4: noisy content (amp 64), the serial index (one wave per packet, k_index_walk) against the
# exact parallel index (k_index_summarize / resolve / emit) at large launches; speculation off for both
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), {k:v['ms'] for k,v in j['kernels'].items()})"; }
for n in 4096 8192 16384; do
  for idx in serial parallel; do
    MI_RTJ_SPEC=0 MI_RTJ_INDEX=$idx timeout -k 10 400 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 4 --warmup 2 --amp 64 --frames $n 2>$O/noisy_$idx_$n.err | pr "amp64 frames=$n index=$idx" | tee -a $O/noisy_serial_ab.txt
  done
done
for amp in 32 40; do
  for idx in serial parallel; do
    MI_RTJ_SPEC=0 MI_RTJ_INDEX=$idx timeout -k 10 400 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 4 --warmup 2 --amp $amp --frames 8192 2>/dev/null | pr "amp$amp frames=8192 index=$idx" | tee -a $O/noisy_serial_ab.txt
  done
done
