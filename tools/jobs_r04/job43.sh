#!/bin/bash
# round 4: the second index of long plans made with the plan: tests of the overlap, the default bench line twice
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_decode_policy.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_dyn2.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_dyn2.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); st=j.get('stress_amp64') or {}; print('$1', round(j['value']), j['roofline']['ms_per_launch'], 'mismatches', j.get('parity_mismatches'), 'stress', st.get('frames_per_s'), st.get('index_overlaps_transform'), 'encoder', round(j['encoder']['frames_per_s']))"; }
for rep in 1 2; do timeout -k 10 500 python bench.py 2>/dev/null | pr "default bench" | tee -a $O/dyn2_bench.txt; done
