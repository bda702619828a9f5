#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_decode_policy.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_policy.log 2>&1; echo "policy pytest rc=$?"; tail -15 $O/pytest_policy.log
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 6 --warmup 3"
timeout -k 10 200 python bench.py $B 2>/dev/null | pr "default amp 8 16384"
for amp in 0 16 32; do
  timeout -k 10 200 python bench.py $B --amp $amp --frames 4096 2>/dev/null | pr "policy amp=$amp 4096" | tee -a $O/policy_content.txt
done
timeout -k 10 200 python bench.py $B --quality 128 --frames 4096 2>/dev/null | pr "policy Q=128 4096" | tee -a $O/policy_content.txt
