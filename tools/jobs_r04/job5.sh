#!/bin/bash
# round 4: DV decoder parity on the GPU, the new RTjpeg cases, then job 4 (roles of k_decode_split alone, other content)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_dv.log 2>&1; echo "dv pytest rc=$?"; tail -15 $O/pytest_dv.log
timeout -k 10 600 python -m pytest tests/test_gpu_overlap.py tests/test_plugin_harness.py -m gpu -x -q > $O/pytest_ov.log 2>&1; echo "overlap+harness pytest rc=$?"; tail -8 $O/pytest_ov.log
bash tools/jobs_r04/job4.sh
