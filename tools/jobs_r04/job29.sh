#!/bin/bash
# round 4: k_dv_decode with the packed transform passes: parity, time, instruction counts
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_dvpk.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_dvpk.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']), 'mismatches', j.get('parity_mismatches'))"; }
: > $O/dv_pk.txt
timeout -k 10 300 python bench.py --config dv --steps 20 --warmup 3 2>/dev/null | pr "packed passes" | tee -a $O/dv_pk.txt
timeout -k 10 300 python bench.py --config dv --amp 0 --steps 20 --warmup 3 2>/dev/null | pr "packed passes, amp 0" | tee -a $O/dv_pk.txt
timeout -k 10 300 python bench.py --config dv --amp 32 --steps 20 --warmup 3 2>/dev/null | pr "packed passes, amp 32" | tee -a $O/dv_pk.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/dvpk_$tag -- python3 bench.py --config dv --no-cpu --steps 2 --warmup 1 > $O/dvpk_$tag.log 2>&1
done
python - <<'PY' | tee -a gpurun_out/r4/dv_pk.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r4/dvpk_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dv_decode" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(x) / len(x) / 138240, 1) for c, x in acc.items()}, "per wave (launch of 1,024 frames = 138,240 waves)")
PY
