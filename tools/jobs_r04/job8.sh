#!/bin/bash
# round 4: DV decoder: debug comparison, then the parity suite, then the DV bench, then the RTjpeg jobs
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
bash tools/jobs_r04/job7.sh > $O/job7.log 2>&1; grep -c "differ at nat" $O/job7.log; grep "blocks with wrong\|pixels differ" $O/job7.log
timeout -k 5 300 python -m pytest tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_dv.log 2>&1; echo "dv pytest rc=$?"; tail -12 $O/pytest_dv.log
timeout -k 5 300 python bench.py --config dv --steps 10 --warmup 3 > $O/bench_dv.json 2> $O/bench_dv.err; echo "dv bench rc=$?"; tail -c 1500 $O/bench_dv.json; tail -3 $O/bench_dv.err
timeout -k 10 900 python -m pytest tests/test_gpu_overlap.py tests/test_plugin_harness.py -m gpu -x -q > $O/pytest_ov.log 2>&1; echo "overlap+harness pytest rc=$?"; tail -8 $O/pytest_ov.log
bash tools/jobs_r04/job4.sh
