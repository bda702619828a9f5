#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_lcg.py tests/test_gpu_decode_policy.py -m gpu -x -q > $O/pytest_lcg.log 2>&1; echo "lcg+policy tests rc=$?"; tail -5 $O/pytest_lcg.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_enc -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 3 --warmup 1 --frames 4096 > $O/trace_enc.log 2>&1
python tools/kstats.py $O/trace_enc k_ | head -12
timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 5 --warmup 2 --content lcg 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lcg content', j['value'], j['config']['avg_packet_bytes'], j['roofline']['ms_per_launch'], j['encoder'])"
