#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (gpurun_out/pmc/*/…counter_collection.csv) per kernel:
average counter value per dispatch.  Usage: python tools/pmc_summary.py gpurun_out/pmc [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
# template instantiations of one kernel (the two walker forms, of which one returns at once): the busiest one
# is listed under the bare name, the others under their full names
groups = defaultdict(list)
for k in acc:
    groups[k.split("<")[0]].append(k)
for base, names in groups.items():
    if names == [base]:
        continue
    weight = lambda k: sum(sum(v) for v in acc[k].values())
    names.sort(key=weight, reverse=True)
    acc[base] = acc.pop(names[0])
out = {}
for k, ctrs in sorted(acc.items()):
    out[k] = {c: sum(v) / len(v) for c, v in sorted(ctrs.items())}
    out[k]["dispatches"] = max(len(v) for v in ctrs.values())
txt = json.dumps(out, indent=1)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
