#!/bin/bash
# round 4: long plans overlap index and transform while the host sees the decode policy's classic mode: parity, the noisy figures
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_decode_policy.py tests/test_gpu_parity.py tests/test_gpu_spec_index.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest_dyn.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_dyn.log
[ $rc -eq 0 ] || { grep -n "Error\|assert" $O/pytest_dyn.log | head; exit 1; }
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); st=j.get('stress_amp64') or {}; print('$1', round(j['value']), j['roofline']['ms_per_launch'], 'mismatches', j.get('parity_mismatches'), 'stress', st.get('frames_per_s'), st.get('index_overlaps_transform'))"; }
timeout -k 10 400 python bench.py --no-e2e --no-sweep 2>/dev/null | pr "bench (cpu legs, stress)" | tee $O/dyn_bench.txt
for amp in 32 40 16; do
timeout -k 10 300 python bench.py --no-cpu --no-e2e --no-sweep --no-stress --content hash --amp $amp --steps 8 --warmup 8 2>/dev/null | pr "amp$amp" | tee -a $O/dyn_bench.txt
MI_RTJ_OVERLAP=0 timeout -k 10 300 python bench.py --no-cpu --no-e2e --no-sweep --no-stress --content hash --amp $amp --steps 8 --warmup 8 2>/dev/null | pr "amp$amp MI_RTJ_OVERLAP=0" | tee -a $O/dyn_bench.txt
done
