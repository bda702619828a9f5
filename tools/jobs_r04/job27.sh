#!/bin/bash
# round 4: k_dv_decode as workgroups of several waves sharing the tables: parity, then waves per workgroup / scratch stride
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_dvwg.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_dvwg.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']), 'mismatches', j.get('parity_mismatches'))"; }
: > $O/dv_wg.txt
timeout -k 10 300 python bench.py --config dv --steps 20 --warmup 3 2>/dev/null | pr "product (4 waves, stride 144)" | tee -a $O/dv_wg.txt
for v in w2_s144 w5_s136 w7_s144 w7_s136; do
MI_DV_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_dv_$v.so timeout -k 10 300 python bench.py --config dv --steps 20 --warmup 3 2>/dev/null | pr "$v" | tee -a $O/dv_wg.txt
done
