#!/bin/bash
# round 4: split form as one workgroup per stripe and frame (L + C waves), plain stores in the chroma waves: parity, time, traffic
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
LE=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
LP=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_plainst.so
MI_RTJ_LIB=$LP MI_RTJ_SPLIT_WG=1 timeout -k 10 600 python -m pytest tests/test_gpu_decode_policy.py tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest_wg.log 2>&1; rc=$?; echo "pytest (workgroup form, plain chroma stores) rc=$rc"; tail -3 $O/pytest_wg.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 8 --warmup 3"
: > $O/wg_ab.txt
for rep in 1 2; do
MI_RTJ_SPLIT=0 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "classic" | tee -a $O/wg_ab.txt
MI_RTJ_LIB=$LE MI_RTJ_SPLIT=1 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split, nt chroma stores" | tee -a $O/wg_ab.txt
MI_RTJ_LIB=$LP MI_RTJ_SPLIT=1 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split, plain chroma stores" | tee -a $O/wg_ab.txt
MI_RTJ_LIB=$LE MI_RTJ_SPLIT=1 MI_RTJ_SPLIT_WG=1 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split as workgroups, nt" | tee -a $O/wg_ab.txt
MI_RTJ_LIB=$LP MI_RTJ_SPLIT=1 MI_RTJ_SPLIT_WG=1 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split as workgroups, plain" | tee -a $O/wg_ab.txt
done
MI_RTJ_LIB=$LP MI_RTJ_SPLIT=1 MI_RTJ_SPLIT_WG=1 MI_RTJ_LUMA_WAVES=6 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split as workgroups, plain, lw=6" | tee -a $O/wg_ab.txt
MI_RTJ_LIB=$LP MI_RTJ_SPLIT=1 MI_RTJ_SPLIT_WG=1 MI_RTJ_LUMA_WAVES=3 timeout -k 10 200 python bench.py $B 2>/dev/null | pr "split as workgroups, plain, lw=3" | tee -a $O/wg_ab.txt
for v in "plain $LP 0" "wgplain $LP 1" "wgnt $LE 1"; do set -- $v
  for set in FETCH_SIZE WRITE_SIZE; do
    if [ $3 -eq 1 ]; then export MI_RTJ_SPLIT_WG=1; else unset MI_RTJ_SPLIT_WG; fi
    MI_RTJ_LIB=$2 MI_RTJ_SPLIT=1 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/wgpmc_$1_$set -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 2 --warmup 1 > $O/wgpmc_$1_$set.log 2>&1
  done
done
python - <<'PY' | tee -a gpurun_out/r4/wg_ab.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r4/wgpmc_*/**/*counter_collection.csv", recursive=True):
    v = re.search(r"wgpmc_([a-z]+)_", f).group(1)
    for r in csv.DictReader(open(f)):
        if "k_decode_split" in r["Kernel_Name"]: acc[v][r["Counter_Name"]].append(float(r["Counter_Value"]))
for v in sorted(acc):
    d = {c: sum(x) / len(x) * 1024 / 1e9 for c, x in acc[v].items()}
    print(v, "fetch (x2)", round(2 * d.get("FETCH_SIZE", 0), 2), "GB, write", round(d.get("WRITE_SIZE", 0), 2), "GB per launch (algorithmic: 9.64 + 51.34)")
PY
