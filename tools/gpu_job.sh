set -u
mkdir -p gpurun_out/r02z
timeout -k 10 400 python bench.py --config streams4k > gpurun_out/r02z/bench_streams4k.json 2> gpurun_out/r02z/streams4k.err; echo "streams4k rc=$?"
timeout -k 10 400 python bench.py --config mixed > gpurun_out/r02z/bench_mixed.json 2> gpurun_out/r02z/mixed.err; echo "mixed rc=$?"
python -c "
import json
for n in ('streams4k','mixed'):
    d=json.load(open('gpurun_out/r02z/bench_%s.json'%n)); print(n, d['value'], d['unit'], d.get('parity_checked'), d.get('parity_mismatches'))"
for i in 1 2 3; do python tools/e2e_bench.py 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:(v['fps'] if isinstance(v,dict) else v) for k,v in d.items() if k!='workload'})"; done | tee gpurun_out/r02z/e2e_3runs.txt
