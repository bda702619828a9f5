#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_pool.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_overlap.py -m gpu -x -q > $O/t_pool.log 2>&1; echo "pool pytest rc=$?"; tail -15 $O/t_pool.log
grep -q " passed" $O/t_pool.log && ! grep -q "failed" $O/t_pool.log || exit 1
bash tools/ab_libs.sh 2 --no-stress --no-e2e --no-sweep --steps 10 -- product lib_pool.so | tee $O/ab_chroma_pool.txt
