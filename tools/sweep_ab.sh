#!/bin/bash
# whole-step frames/s by launch size, with and without the index of the next launch overlapping the transform of this one
cd "$(dirname "$0")/.."
for ov in 0 1; do for n in 256 1024 4096 16384; do
MI_RTJ_OVERLAP=$ov python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 10 --warmup 3 --frames $n 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('overlap $ov frames $n', d['value'], {a:b['ms'] for a,b in d['kernels'].items() if b['ms']>0.02})"
done; done
