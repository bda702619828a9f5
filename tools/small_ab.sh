#!/bin/bash
cd "$(dirname "$0")/.."
for n in 256 1024; do for ro in 0 1; do
MI_RTJ_ROTATE=$ro python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 10 --warmup 3 --frames $n 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('frames $n rotate $ro', d['value'], {a:b['ms'] for a,b in d['kernels'].items() if b['ms']>0.02})"
done; done
