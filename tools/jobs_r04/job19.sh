#!/bin/bash
# round 4: walk_packet (the serial walker rewritten for scalar issue), luma/chroma waves per XCD of the split form
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
timeout -k 10 900 python -m pytest tests/test_gpu_decode_policy.py tests/test_gpu_parity.py tests/test_gpu_spec_index.py -m gpu -x -q > $O/pytest_walk.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_walk.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']), j.get('stress_amp64'))"; }
B="--no-cpu --no-e2e --no-sweep"
: > $O/walk_ab.txt
timeout -k 10 300 python bench.py $B --steps 6 --warmup 3 2>/dev/null | pr "headline+stress" | tee -a $O/walk_ab.txt
timeout -k 10 300 python bench.py $B --no-stress --content hash --amp 32 --steps 6 --warmup 6 2>/dev/null | pr "amp32" | tee -a $O/walk_ab.txt
MI_RTJ_SERIAL_MIN=1 timeout -k 10 300 python bench.py $B --no-stress --content hash --amp 32 --steps 6 --warmup 6 2>/dev/null | pr "amp32 serial_min=1" | tee -a $O/walk_ab.txt
MI_RTJ_SPEC=0 MI_RTJ_INDEX=serial timeout -k 10 300 python bench.py $B --no-stress --content hash --amp 32 --steps 6 --warmup 3 2>/dev/null | pr "amp32 index=serial" | tee -a $O/walk_ab.txt
MI_RTJ_SPEC=0 MI_RTJ_INDEX=serial timeout -k 10 300 python bench.py $B --no-stress --content hash --amp 64 --steps 6 --warmup 3 2>/dev/null | pr "amp64 index=serial" | tee -a $O/walk_ab.txt
MI_RTJ_SPEC=0 MI_RTJ_INDEX=serial timeout -k 10 300 python bench.py $B --no-stress --steps 6 --warmup 3 2>/dev/null | pr "headline content index=serial" | tee -a $O/walk_ab.txt
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3"
: > $O/lwcw_ab.txt
MI_RTJ_SPLIT=0 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "classic" | tee -a $O/lwcw_ab.txt
for cw in 1 2; do for lw in 2 4 7 8 9 10 12 13; do
  MI_RTJ_LIB=$L MI_RTJ_SPLIT=1 MI_RTJ_LUMA_WAVES=$lw MI_RTJ_CHROMA_WAVES=$cw timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split lw=$lw cw=$cw" | tee -a $O/lwcw_ab.txt
done; done
MI_RTJ_LIB=$L MI_RTJ_SPLIT=1 MI_RTJ_LUMA_WAVES=4 MI_RTJ_XCD_ROT=257 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split lw=4 cw=1 chroma first" | tee -a $O/lwcw_ab.txt
MI_RTJ_LIB=$L MI_RTJ_SPLIT=1 MI_RTJ_LUMA_WAVES=6 MI_RTJ_XCD_ROT=257 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split lw=6 cw=1 chroma first" | tee -a $O/lwcw_ab.txt
MI_RTJ_SPLIT=0 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "classic" | tee -a $O/lwcw_ab.txt
