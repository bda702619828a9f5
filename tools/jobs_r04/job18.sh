#!/bin/bash
# round 4: the XCD-aware split form with stripes that move on from frame to frame; luma waves per XCD
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
timeout -k 10 600 python -m pytest tests/test_gpu_decode_policy.py tests/test_gpu_parity.py tests/test_gpu_dc_only.py -m gpu -x -q > $O/pytest_xrot.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_xrot.log
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3"
: > $O/xrot_ab.txt
MI_RTJ_SPLIT=0 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "classic" | tee -a $O/xrot_ab.txt
for rot in 0 1 3 5; do MI_RTJ_LIB=$L MI_RTJ_SPLIT=1 MI_RTJ_XCD_ROT=$rot timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split rot=$rot lw=6" | tee -a $O/xrot_ab.txt; done
for lw in 3 4 5 8 11; do MI_RTJ_LIB=$L MI_RTJ_SPLIT=1 MI_RTJ_XCD_ROT=1 MI_RTJ_LUMA_WAVES=$lw timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split rot=1 lw=$lw" | tee -a $O/xrot_ab.txt; done
MI_RTJ_SPLIT=0 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "classic" | tee -a $O/xrot_ab.txt
MI_RTJ_SPLIT=1 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split product (rot=1 lw=6)" | tee -a $O/xrot_ab.txt
