#!/bin/bash
# round 4: the XCD-aware split form: parity, speed against the classic form on one box, HBM traffic
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_decode_policy.py tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_spec_index.py tests/test_gpu_overlap.py tests/test_gpu_configs.py tests/test_gpu_dv.py -m gpu -x -q > $O/pytest_xcd.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_xcd.log
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3"
for sp in 1 0 1 0; do MI_RTJ_SPLIT=$sp timeout -k 10 200 python bench.py $B 2>/dev/null | pr "xcd-aware split=$sp" | tee -a $O/xcd_ab.txt; done
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/xcd_pmc_$set -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 2 --warmup 1 > $O/xcd_pmc_$set.log 2>&1
done
python - <<'PY' | tee -a gpurun_out/r4/xcd_ab.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r4/xcd_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
        if k.startswith("k_decode_split") or k.startswith("k_spec_walk"): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x) * 1024 / 1e9, 2) for c, x in v.items()}, "GB raw per launch (FETCH x2 for the decode kernel, x1.1 for the walker)")
PY
