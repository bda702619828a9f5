#!/bin/bash
# round 4: where +-32 content spends its time by index mode and decode form; parity of the final split launch shape
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), {k: v['ms'] for k, v in j['kernels'].items()}, j.get('decode_form'), j['speculative_index'])"; }
B="--no-cpu --no-e2e --no-sweep --no-stress --content hash"
: > $O/amp32_modes.txt
for amp in 16 24 32 40 48; do
MI_RTJ_SPLIT=0 timeout -k 10 300 python bench.py $B --amp $amp --steps 6 --warmup 6 2>/dev/null | pr "amp$amp classic-form spec" | tee -a $O/amp32_modes.txt
MI_RTJ_SPLIT=0 MI_RTJ_SPEC=0 MI_RTJ_INDEX=serial timeout -k 10 300 python bench.py $B --amp $amp --steps 6 --warmup 3 2>/dev/null | pr "amp$amp classic-form serial" | tee -a $O/amp32_modes.txt
done
timeout -k 10 300 python bench.py $B --amp 32 --steps 6 --warmup 6 2>/dev/null | pr "amp32 policy spec" | tee -a $O/amp32_modes.txt
timeout -k 10 300 python bench.py $B --amp 32 --steps 6 --warmup 70 2>/dev/null | pr "amp32 policy spec warm70" | tee -a $O/amp32_modes.txt
timeout -k 10 900 python -m pytest tests/test_gpu_decode_policy.py tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest_lw.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_lw.log
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3"
for sp in 1 0 1 0; do MI_RTJ_SPLIT=$sp timeout -k 10 200 python bench.py $B 2>/dev/null | pr "final split=$sp" | cut -c1-200 | tee -a $O/amp32_modes.txt; done
