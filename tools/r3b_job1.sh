#!/bin/bash
# round-3 session-2 measurement job 1: tests of the spec path, by_batch sweep, timeline of 1024-picture launches, what bounds a 4K session
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_spec_index.py tests/test_gpu_parity.py tests/test_gpu_overlap.py -m gpu -x -q > $O/t1.log 2>&1; echo "pytest rc=$?" | tee -a $O/t1.log
grep -q "rc=0" $O/t1.log || { tail -30 $O/t1.log; exit 1; }
timeout -k 10 300 python bench.py --no-stress --no-e2e > $O/bench_walk_any.json 2> $O/bench_walk_any.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3b/bench_walk_any.json').read().strip().splitlines()[-1])
print(d['value'], {k:v['ms'] for k,v in d['kernels'].items()})
print({k:(v['frames_per_s'] if isinstance(v,dict) else v) for k,v in d['by_batch'].items() if k!='note'})
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl1024 -- python3 bench.py --frames 1024 --steps 30 --warmup 5 --no-cpu --no-stress --no-e2e --no-sweep > $O/tl1024.log 2>&1; echo "trace rc=$?"
python tools/timeline.py $O/tl1024 8 | tee $O/timeline_1024.txt
rm -rf $O/tl1024
for sk in 0 1 2; do
  MI_RTJ_EXP_SKIP=$sk W=3840 H=2160 PK=24 REP=8 DEPTH=6 python - "4K skip=$sk" <<'PY' | tee -a $O/e2e_4k_bounds.txt
import sys, json, os
sys.path.insert(0, '.')
import tools.e2e_bench as E
r = E.run(int(os.environ["W"]), int(os.environ["H"]), packets=int(os.environ["PK"]), repeat=int(os.environ["REP"]), depth=int(os.environ["DEPTH"]), flavours=("_pipe",), two_streams=False)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:40s} {one.get('fps')}")
PY
done
