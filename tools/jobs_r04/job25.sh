#!/bin/bash
# round 4: why the chroma waves write 22.8 GB for 17.1 GB of chroma planes: row alignment (width 2048: chroma rows of 1024 B) and the store kind
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
LE=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_exp.so
LP=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg_plainst.so
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', j['roofline']['ms_per_launch'], round(j['value']))"; }
: > $O/chroma_store_ab.txt
for W in 1920 2048; do for lib in $LE $LP; do
  n=$(basename $lib .so)
  MI_RTJ_LIB=$lib MI_RTJ_SPLIT=1 timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --width $W --frames 8192 --steps 8 --warmup 3 2>/dev/null | pr "width=$W $n both roles" | tee -a $O/chroma_store_ab.txt
  MI_RTJ_LIB=$lib MI_RTJ_SPLIT=1 MI_RTJ_SPLIT_ONLY=2 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cst_${W}_$n -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --width $W --frames 8192 --steps 2 --warmup 1 > $O/cst_${W}_$n.log 2>&1
done; done
python - <<'PY' | tee -a gpurun_out/r4/chroma_store_ab.txt
import csv, glob, collections, re
for d in sorted(glob.glob("gpurun_out/r4/cst_*")):
    if not d.endswith(".log"):
        v = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_decode_split" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE": v.append(float(r["Counter_Value"]))
        W = int(re.search(r"cst_(\d+)_", d).group(1))
        exp = 8192 * W * 1088 / 2
        if v: print(d.split("/")[-1], "chroma waves only: WRITE_SIZE", round(sum(v) / len(v) * 1024 / 1e9, 2), "GB for", round(exp / 1e9, 2), "GB of chroma planes")
PY
