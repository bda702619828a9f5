set -u
mkdir -p gpurun_out/r02o
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02o/pytest.log; tail -3 gpurun_out/r02o/pytest.log)
python tools/e2e_bench.py 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print({k:(v['fps'] if isinstance(v,dict) else v) for k,v in d.items() if k!='workload'})"
