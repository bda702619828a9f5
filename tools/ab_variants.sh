#!/bin/bash
# Builds compile-time variants of libmi_rtjpeg.so into gpurun_out/ab/ (run HERE, hipcc cross-compiles)
# and, on the GPU box, benches each: bash tools/ab_variants.sh build | run
set -u
cd "$(dirname "$0")/.."
mkdir -p gmerlin-avdecoder_amd/lib/ab
declare -A V=( [base]="" )
if [ "${1:-build}" = build ]; then
  for k in "${!V[@]}"; do
    MI_RTJ_CFLAGS="${V[$k]}" python -c "
import importlib,sys; sys.path.insert(0,'.')
b=importlib.import_module('gmerlin-avdecoder_amd.build'); print(b.build(force=True,out='gmerlin-avdecoder_amd/lib/ab/lib_$k.so'))"
  done
else
  for rep in 1 2; do for k in "${!V[@]}"; do
    MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_$k.so timeout -k 10 120 python bench.py --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$k', d['value'], {a:b['ms'] for a,b in d['kernels'].items()})"
  done; done
fi
