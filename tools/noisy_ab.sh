#!/bin/bash
# exact index on noisy content: bash tools/noisy_ab.sh > gpurun_out/<tag>/noisy.txt
cd "$(dirname "$0")/.."
run() { local label=$1; shift; env "$@" python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 5 --warmup 2 --frames 1024 --amp ${AMP:-64} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], {a:b['ms'] for a,b in d['kernels'].items() if b['ms']>0.02})"; }
for amp in 64 32 16; do
run "amp $amp exact index, table emit" MI_RTJ_SPEC=0 AMP=$amp
done
for amp in 64 32 16; do
run "amp $amp exact index, walk emit" MI_RTJ_SPEC=0 MI_RTJ_EMIT=walk AMP=$amp
done
run "amp 8 exact index, table emit" MI_RTJ_SPEC=0 AMP=8
run "amp 8 exact index, walk emit" MI_RTJ_SPEC=0 MI_RTJ_EMIT=walk AMP=8
