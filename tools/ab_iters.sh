#!/bin/bash
# k_decode groups per wave (MIRTJ_DEC_ITERS): build variants HERE (bash tools/ab_iters.sh build), bench them on the GPU box (run)
set -u
cd "$(dirname "$0")/.."
mkdir -p gmerlin-avdecoder_amd/lib/ab
ITERS="${ITERS:-8 11}"
if [ "${1:-build}" = build ]; then
  for k in $ITERS; do
    MI_RTJ_CFLAGS="-DMIRTJ_DEC_ITERS=$k" python -c "
import importlib,sys; sys.path.insert(0,'.')
b=importlib.import_module('gmerlin-avdecoder_amd.build'); print(b.build(force=True,out='gmerlin-avdecoder_amd/lib/ab/lib_i$k.so'))"
  done
else
  for rep in 1 2; do for k in $ITERS; do for cfg in "" "--width 3840 --height 2160 --frames 512" "--quality 128" "--width 1280 --height 720"; do
    MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_i$k.so timeout -k 10 120 python bench.py --no-cpu $cfg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('i$k', '[$cfg]', d['value'], d['kernels']['k_decode']['ms'])"
  done; done; done
fi
