#!/usr/bin/env python3
"""Where the time of a step goes BETWEEN kernels: reads a rocprofv3 --kernel-trace CSV (found under argv[1]) and
prints, for the steady-state steps of a bench run (from one k_decode end to the next), the busy time of every kernel,
the union of all kernel intervals (device busy) and what is left (no kernel running: dispatch gaps, cross-stream waits).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --frames 1024 ...
    python tools/timeline.py gpurun_out/tl [skip_steps]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("mirtj::", "").replace("void ", "")
    return n.split("<")[0]


def main():
    root = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    paths = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        print("no kernel_trace.csv under", root)
        return 1
    rows = []
    for r in csv.DictReader(open(paths[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    dec_ends = [e for s, e, n in rows if n == "k_decode"]
    if len(dec_ends) < skip + 3:
        print("too few k_decode launches:", len(dec_ends))
        return 1
    # steady state: the longest run of steps of similar length (the timed loop), after `skip`
    steps = list(zip(dec_ends[skip:-1], dec_ends[skip + 1:]))
    lens = sorted(b - a for a, b in steps)
    med = lens[len(lens) // 2]
    steps = [(a, b) for a, b in steps if 0.7 * med < b - a < 1.3 * med]
    busy = defaultdict(float)
    calls = defaultdict(int)
    union = 0.0
    total = 0.0
    for a, b in steps:
        total += b - a
        iv = []
        for s, e, n in rows:
            if e <= a or s >= b:
                continue
            s2, e2 = max(s, a), min(e, b)
            busy[n] += e2 - s2
            calls[n] += 1
            iv.append((s2, e2))
        iv.sort()
        cur_s, cur_e = None, None
        for s, e in iv:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        if cur_e is not None:
            union += cur_e - cur_s
    n = len(steps)
    print(f"steps {n}  step {total / n / 1e3:.1f} us  some kernel running {union / n / 1e3:.1f} us  none {(total - union) / n / 1e3:.1f} us")
    for k in sorted(busy, key=lambda k: -busy[k]):
        print(f"  {k:26s} {busy[k] / n / 1e3:9.1f} us per step  ({calls[k] / n:.1f} dispatches)")
    print(f"  sum of kernel times {sum(busy.values()) / n / 1e3:.1f} us (kernels of two streams overlap)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
