#!/bin/bash
# Dynamic instruction census of k_decode: rocprofv3 --pmc SQ_INSTS_VALU / SALU / LDS of the
# product and of the census builds (no range and shape tests; no transform; neither transform nor parse), 4096 pictures per launch.
#   build the libraries here (see the list below), then on the GPU box: bash tools/census.sh > gpurun_out/<tag>/census.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/census
rm -rf $OUT; mkdir -p $OUT
for k in product c_nopk c_notransform c_neither; do  # (parse = c_notransform - c_neither)
  if [ $k = product ]; then export MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else export MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_$k.so; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/$k -- python3 bench.py --frames 4096 --steps 2 --warmup 1 --no-cpu --no-stress --no-e2e --no-sweep > $OUT/$k.log 2>&1
  python3 - $OUT/$k $k <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
        if k.startswith("k_decode"):
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(f"{sys.argv[2]:14s} {k:24s} " + "  ".join(f"{n} {sum(v)/len(v)/1e6:9.1f} M" for n, v in sorted(c.items())))
PY
done
