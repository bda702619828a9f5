#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
MI_RTJ_LIB=$L timeout -k 5 200 python tools/pool_stamps.py 4096 > $O/pool_stamps.txt 2>&1; cat $O/pool_stamps.txt
MI_RTJ_LIB=$L MI_RTJ_SPLIT_ONLY=2 timeout -k 5 200 python tools/pool_stamps.py 4096 > $O/pool_stamps_only.txt 2>&1; cat $O/pool_stamps_only.txt
timeout -k 5 300 python -m pytest tests/test_gpu_dv.py -m gpu -x -q 2>&1 | tail -3
bash tools/jobs_r04/job9.sh
