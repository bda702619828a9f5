#!/bin/bash
# k_decode waves per part (MI_RTJ_DEC_SLOTS) at the bench's default workload, on the GPU box: bash tools/ab_slots.sh
set -u
cd "$(dirname "$0")/.."
for rep in 1 2; do for s in ${SLOTS:-24 25 26 27 29 30 31 33 35 37 39 43 45 51}; do
  MI_RTJ_DEC_SLOTS=$s timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('slots $s', d['value'], d['kernels']['k_decode']['ms'])"
done; done
