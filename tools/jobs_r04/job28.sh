#!/bin/bash
# round 4: k_dv_decode's transform section: arithmetic against stores (timing builds)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_dv1.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_dv1.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
: > $O/dv_sections2.txt
timeout -k 10 300 python bench.py --config dv --no-cpu --steps 20 --warmup 3 2>/dev/null | pr "full (1 wave per workgroup)" | tee -a $O/dv_sections2.txt
for k in 1 2 3 8 16 24; do
MI_DV_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_dv_skip$k.so timeout -k 10 300 python bench.py --config dv --no-cpu --steps 20 --warmup 3 2>/dev/null | pr "skip=$k" | tee -a $O/dv_sections2.txt
done
for set in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/dvpmc_$tag -- python3 bench.py --config dv --no-cpu --steps 2 --warmup 1 > $O/dvpmc_$tag.log 2>&1
done
python - <<'PY' | tee -a gpurun_out/r4/dv_sections2.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r4/dvpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_dv_decode" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(x) / len(x) / 1e6, 2) for c, x in acc.items()}, "millions per launch of 1,024 frames (138,240 waves)")
PY
