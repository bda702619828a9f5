#!/bin/bash
# round 4: cycles and clock of the two forms of the transform kernel on ONE box (GRBM_GUI_ACTIVE per kernel + durations)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
for sp in 1 0 1 0; do
  d=$O/clk_split${sp}_$RANDOM
  MI_RTJ_SPLIT=$sp timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 4 --warmup 2 > $d.log 2>&1
  python - "$d" "$sp" <<'PY' | tee -a gpurun_out/r4/clock_by_form.txt
import csv, glob, sys, collections
d, sp = sys.argv[1], sys.argv[2]
cyc = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[k].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in cyc:
    if not k.startswith("k_decode") and not k.startswith("k_spec_walk"): continue
    c = sum(cyc[k]) / len(cyc[k]) / 8.0; t = sum(dur[k]) / max(len(dur[k]), 1)
    if t > 0.5: print(f"MI_RTJ_SPLIT={sp} {k:24s} {c/1e6:8.2f} M cycles per XCD  {t:7.3f} ms  -> {c/t/1e6:5.3f} GHz")
PY
done
