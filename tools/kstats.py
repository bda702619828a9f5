#!/usr/bin/env python3
"""Prints (kernel, calls, average µs) from a rocprofv3 kernel_stats.csv found under a directory; kernels matching argv[2]."""
import csv, glob, os, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
        if pat in name:
            print(f"{name:28s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  min {float(r['MinNs'])/1e3:10.1f}")
