#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
export MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_pool.so
MI_RTJ_ROTATE=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_overlap.py tests/test_gpu_spec_index.py -m gpu -x -q > $O/t_pool3.log 2>&1; echo "pool+rotate pytest rc=$?"; tail -6 $O/t_pool3.log
rm -rf $O/pmc_pool3
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/pmc_pool3 -- python3 bench.py --frames 4096 --steps 3 --warmup 1 --no-cpu --no-stress --no-e2e --no-sweep > $O/pmc_pool3.log 2>&1; echo "pmc rc=$?"
python - $O/pmc_pool3 pool3 <<'PY' | tee -a $O/pmc_chroma_pool3.txt
import csv, glob, os, sys, collections
root, label = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_decode" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print(label, k, {c: round(x / n[(k, c)] / 1e6, 1) for c, x in v.items()})
PY
rm -rf $O/pmc_pool3
unset MI_RTJ_LIB
bash tools/ab_libs.sh 2 --no-stress --no-e2e --no-sweep --steps 8 -- product lib_pool.so | tee $O/ab_chroma_pool3.txt
