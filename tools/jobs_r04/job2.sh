#!/bin/bash
# round 4: where does the split decode's time go?  kernel trace, luma-only / chroma-only timing builds, VALU count
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 6 --warmup 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -- python3 bench.py $B > $O/trace2.log 2>&1 || exit 1
python tools/kstats.py $O/trace2 k_ > $O/kstats2.txt; cat $O/kstats2.txt
L=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_exp.so
for only in 1 2; do
  MI_RTJ_LIB=$L MI_RTJ_SPLIT_ONLY=$only timeout -k 10 200 python bench.py $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('only=$only', j['roofline']['ms_per_launch'], j['value'])" | tee -a $O/split_only.txt
done
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/pmc2 -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 2 --warmup 1 > $O/pmc2.log 2>&1 || exit 1
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("gpurun_out/r4/pmc2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_decode" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"]=="SQ_WAVES": n[k]+=1
for k in acc:
    print(k, n[k], {c: v/max(n[k],1) for c,v in acc[k].items()})
PY
