set -u
mkdir -p gpurun_out/r02e
timeout -k 10 500 python bench.py > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r02e/bench.json')); print(d['value'], d['roofline'], d.get('roofline_valu',{}).get('frac_of_expensive_rate'), d['parity_checked'], d['parity_mismatches'], d['end_to_end']['fps'])"
