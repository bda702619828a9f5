#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
export MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_pool.so
MI_RTJ_ROTATE=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_overlap.py tests/test_gpu_spec_index.py -m gpu -x -q > $O/t_pool_rot.log 2>&1; echo "pool+rotate pytest rc=$?"; tail -12 $O/t_pool_rot.log
timeout -k 10 300 python bench.py --no-stress --no-e2e --no-sweep --steps 5 > $O/bench_pool.json 2> $O/bench_pool.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_pool.json').read().strip().splitlines()[-1]); print(d['value'], d['parity_checked'], d['parity_mismatches'], d['kernels']['k_decode']['ms'])"
