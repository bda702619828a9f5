#!/bin/bash
# round 4: who writes / fetches what in the split form (PMC on timing builds with one role launched)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
for role in 0 1 2; do for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $set | tr ' ' '_')
  if [ $role -eq 0 ]; then export -n MI_RTJ_SPLIT_ONLY; unset MI_RTJ_SPLIT_ONLY; else export MI_RTJ_SPLIT_ONLY=$role; fi
  MI_RTJ_LIB=$L MI_RTJ_SPLIT=1 timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/roles_pmc_${role}_$tag -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 2 --warmup 1 > $O/roles_pmc_${role}_$tag.log 2>&1
done; done
python - <<'PY' | tee gpurun_out/r4/roles_traffic.txt
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r4/roles_pmc_*/**/*counter_collection.csv", recursive=True):
    role = re.search(r"roles_pmc_(\d)_", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
        if k.startswith("k_decode_split"): acc[role][r["Counter_Name"]].append(float(r["Counter_Value"]))
for role in sorted(acc):
    print({"0": "both roles", "1": "luma waves only", "2": "chroma waves only"}[role], {c: round(sum(x) / len(x) / 1e6, 3) for c, x in acc[role].items()}, "(millions per launch; FETCH_SIZE / WRITE_SIZE in KiB units)")
PY
