#!/bin/bash
# round 4: k_dv_decode by section (timing builds with sections compiled out: wrong pictures)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
: > $O/dv_sections.txt
timeout -k 10 300 python bench.py --config dv --no-cpu --steps 20 --warmup 3 2>/dev/null | pr "full" | tee -a $O/dv_sections.txt
for k in 1 2 3 4 7; do
MI_DV_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_dv_skip$k.so timeout -k 10 300 python bench.py --config dv --no-cpu --steps 20 --warmup 3 2>/dev/null | pr "skip=$k (1 no pass 2/3, 2 no transforms+stores, 4 no pass 1)" | tee -a $O/dv_sections.txt
done
