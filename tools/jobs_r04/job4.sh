#!/bin/bash
# round 4: the two roles of k_decode_split alone (timing build, wrong pictures), and the split on other content
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 6 --warmup 2"
L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']), j.get('kernels_ms_per_step'))"; }
for only in 1 2; do
  MI_RTJ_LIB=$L MI_RTJ_SPLIT_ONLY=$only timeout -k 10 200 python bench.py $B 2> $O/only$only.err | pr only=$only | tee -a $O/split_only.txt
done
for amp in 0 16 32; do
  for sp in 1 0; do
    MI_RTJ_SPLIT=$sp timeout -k 10 200 python bench.py $B --amp $amp --frames 4096 2>/dev/null | pr "amp=$amp split=$sp" | tee -a $O/split_content.txt
  done
done
for q in 128 64; do
  for sp in 1 0; do
    MI_RTJ_SPLIT=$sp timeout -k 10 200 python bench.py $B --quality $q --frames 4096 2>/dev/null | pr "Q=$q split=$sp" | tee -a $O/split_content.txt
  done
done
