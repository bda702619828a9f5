set -u
mkdir -p gpurun_out/r02o
python tools/pcie_probe.py 2>/dev/null | tee gpurun_out/r02o/pcie_probe.json
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02o/pytest.log; tail -5 gpurun_out/r02o/pytest.log)
( time timeout -k 10 600 python bench.py > gpurun_out/r02o/bench_final.json 2> gpurun_out/r02o/bench.err ) 2>&1 | tail -4
python -c "
import json; d=json.loads(open('gpurun_out/r02o/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d.get('roofline_valu',{}).get('frac'), d['parity_checked'], d['parity_mismatches'], d['end_to_end'].get('fps'))"
