#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_quarters.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dc_only.py -m gpu -x -q > $O/t_quarters.log 2>&1; echo "quarters pytest rc=$?"; tail -2 $O/t_quarters.log
bash tools/ab_libs.sh 3 --no-stress --no-e2e --no-sweep --steps 10 -- product lib_quarters.so | tee $O/ab_parse_quarters.txt
