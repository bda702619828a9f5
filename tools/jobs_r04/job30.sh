#!/bin/bash
# round 4: k_dv_decode with passes 2 and 3 of a workgroup done by one wave (a lane per macroblock / segment)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_dvb.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_dvb.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']), 'mismatches', j.get('parity_mismatches'))"; }
: > $O/dv_b.txt
timeout -k 10 300 python bench.py --config dv --steps 20 --warmup 3 2>/dev/null | pr "6 waves per workgroup" | tee -a $O/dv_b.txt
for w in 1 2 3 4; do
MI_DV_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_dv_b$w.so timeout -k 10 300 python bench.py --config dv --steps 20 --warmup 3 2>/dev/null | pr "$w waves per workgroup" | tee -a $O/dv_b.txt
done
timeout -k 10 300 python bench.py --config dv --amp 0 --steps 20 --warmup 3 2>/dev/null | pr "6 waves, amp 0" | tee -a $O/dv_b.txt
timeout -k 10 300 python bench.py --config dv --amp 32 --steps 20 --warmup 3 2>/dev/null | pr "6 waves, amp 32" | tee -a $O/dv_b.txt
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/dvb_$tag -- python3 bench.py --config dv --no-cpu --steps 2 --warmup 1 > $O/dvb_$tag.log 2>&1
done
python - <<'PY' | tee -a gpurun_out/r4/dv_b.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r4/dvb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dv_decode" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(x) / len(x) / 138240, 1) for c, x in acc.items()}, "per wave (launch of 1,024 frames = 138,240 waves)")
PY
