set -u
mkdir -p gpurun_out/r02n
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02n/pytest.log; tail -4 gpurun_out/r02n/pytest.log)
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/r02n/bench.json 2>gpurun_out/r02n/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02n/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_valu']['frac_of_expensive_rate'], d['parity_checked'], d['parity_mismatches'], d['end_to_end']['fps'], {k:v['ms'] for k,v in d['kernels'].items() if v['ms']>0.05})"
