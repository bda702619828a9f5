#!/bin/bash
# A/B of prebuilt libraries on the GPU box: bash tools/ab_libs.sh <reps> <bench args...> -- lib_a.so lib_b.so ...
# (libraries relative to gmerlin-avdecoder_amd/lib/ab/; "product" = the in-tree libmi_rtjpeg.so)
set -u
cd "$(dirname "$0")/.."
REPS=$1; shift
ARGS=()
while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for rep in $(seq $REPS); do for k in "$@"; do
  if [ "$k" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/$k; fi
  MI_RTJ_LIB=$L timeout -k 10 180 python bench.py --no-cpu "${ARGS[@]}" 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['value'], {a:b['ms'] for a,b in d['kernels'].items() if b['ms']>0.05})"
done; done
