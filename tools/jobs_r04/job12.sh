#!/bin/bash
# round 4: the whole GPU suite; the walker-traffic microbench under PMC; noisy content with the serial-walker policy
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "full gpu suite rc=$?"; tail -6 $O/pytest_all.log
U=tools/ubench/lane_line_fetch
[ -x $U ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $U $U.hip 2>/dev/null
$U 4096 > $O/lane_line_fetch.txt 2>&1; cat $O/lane_line_fetch.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/llf_pmc -- $U 4096 > $O/llf_pmc.log 2>&1
python - <<'PY' | tee gpurun_out/r4/lane_line_fetch_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r4/llf_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:40s} FETCH_SIZE per launch (KiB, raw): {[round(x) for x in v]}  = {[round(x * 1024 / 2**20) for x in v]} MiB raw, x2 = {[round(x * 2048 / 2**20) for x in v]} MiB (buffer: 4096 MiB)")
PY
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), {k:v['ms'] for k,v in j['kernels'].items()})"; }
for n in 1024 4096 16384; do
  timeout -k 10 400 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 5 --warmup 6 --amp 64 --frames $n 2>/dev/null | pr "amp64 frames=$n policy" | tee -a $O/noisy_policy.txt
done
