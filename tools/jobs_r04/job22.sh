#!/bin/bash
# round 4: the index policy with the rewritten serial walker (no longest lead for launches the walker takes), then DV sections
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_spec_index.py tests/test_gpu_decode_policy.py tests/test_gpu_overlap.py -m gpu -x -q > $O/pytest_pol.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_pol.log
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), {k: v['ms'] for k, v in j['kernels'].items()}, j['speculative_index'])"; }
B="--no-cpu --no-e2e --no-sweep --no-stress --content hash"
: > $O/amp_policy2.txt
for amp in 8 24 32 40; do
timeout -k 10 300 python bench.py $B --amp $amp --steps 6 --warmup 6 2>/dev/null | pr "amp$amp default" | cut -c1-400 | tee -a $O/amp_policy2.txt
done
bash tools/jobs_r04/job21.sh
