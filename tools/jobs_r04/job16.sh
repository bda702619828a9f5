#!/bin/bash
# round 4: power and clock while each form of the transform kernel runs (rocm-smi sampled every 0.5 s over a whole bench run)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O; rm -f $O/smi_samples.txt
for sp in 0 1; do
  MI_RTJ_SPLIT=$sp timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 400 --warmup 2 > $O/smi_bench_$sp.json 2>/dev/null &
  BP=$!
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | python -c "
import sys,re
t=sys.stdin.read()
p=re.search(r'Power \(W\): ([0-9.]+)',t); s=re.search(r'sclk clock level: \S+ \((\d+)Mhz\)',t); m=re.search(r'mclk clock level: \S+ \((\d+)Mhz\)',t)
print('split=$sp power_W', p.group(1) if p else '?', 'sclk_MHz', s.group(1) if s else '?', 'mclk_MHz', m.group(1) if m else '?')" >> $O/smi_samples.txt
    sleep 0.4
  done
  wait $BP
  python -c "
import json; j=json.loads(open('$O/smi_bench_$sp.json').read().strip().split('\n')[-1]); print('split=$sp fps', j['value'], 'ms', j['roofline']['ms_per_launch'])" | tee -a $O/smi_samples.txt
done
python - <<'PY'
import re
rows = [l.split() for l in open("gpurun_out/r4/smi_samples.txt") if "power_W" in l]
for sp in ("split=0", "split=1"):
    r = [(float(x[2]), int(x[4])) for x in rows if x[0] == sp and x[2] != "?"]
    busy = sorted(r, key=lambda v: -v[0])[:12]
    print(sp, "samples", len(r), "top-12 by power:", busy)
PY
