#!/bin/bash
# k_decode waves per frame (MI_RTJ_DEC_SLOTS) by launch size: does a count that fills whole rounds of resident waves help?
cd "$(dirname "$0")/.."
for n in 256 1024 2048; do for s in 0 24 25 26 27 28 30 32 36 43 48 51 64; do
MI_RTJ_OVERLAP=0 MI_RTJ_DEC_SLOTS=$s python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 10 --warmup 3 --frames $n 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('frames $n slots $s', d['value'], d['kernels']['k_decode']['ms'])"
done; done
