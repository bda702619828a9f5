#!/bin/bash
# round 4: DV after the look-up / multiplier trims; from how many packets on the serial walker beats the exact kernels
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_dvc.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_dvc.log
[ $rc -eq 0 ] || exit 1
prd() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']), 'mismatches', j.get('parity_mismatches'))"; }
timeout -k 10 300 python bench.py --config dv --steps 20 --warmup 3 2>/dev/null | prd "dv" | tee $O/dv_c.txt
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), {k: v['ms'] for k, v in j['kernels'].items() if v['ms'] > 0.05})"; }
: > $O/serial_min.txt
for amp in 64 32; do for n in 512 1024 2048; do for sm in 0 256; do
MI_RTJ_SERIAL_MIN=$sm timeout -k 10 300 python bench.py --no-cpu --no-e2e --no-sweep --no-stress --content hash --amp $amp --frames $n --steps 30 --warmup 8 2>/dev/null | pr "amp$amp frames=$n serial_min=$sm" | tee -a $O/serial_min.txt
done; done; done
