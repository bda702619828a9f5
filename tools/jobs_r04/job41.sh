#!/bin/bash
# round 4: the classic form's kernel is not enqueued while the host has seen the policy in its split mode: parity, the bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_decode_policy.py tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_overlap.py tests/test_gpu_configs.py tests/test_plugin_harness.py -m gpu -x -q > $O/pytest_seen.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_seen.log
[ $rc -eq 0 ] || exit 1
pr() { python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(j['value']), j['roofline']['ms_per_launch'], 'mismatches', j.get('parity_mismatches'), (j.get('stress_amp64') or {}).get('frames_per_s'))"; }
timeout -k 10 400 python bench.py --no-e2e --no-sweep 2>/dev/null | pr "bench (cpu legs, stress)" | tee $O/seen_bench.txt
timeout -k 10 300 python bench.py --no-cpu --no-e2e --no-sweep --no-stress --content hash --amp 32 --steps 6 --warmup 6 2>/dev/null | pr "amp32" | tee -a $O/seen_bench.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seen_trace -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 10 > $O/seen_trace.log 2>&1
python - <<'PY' | tee -a gpurun_out/r4/seen_bench.txt
import csv, glob
for f in glob.glob("gpurun_out/r4/seen_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("mirtj::", "").replace("void ", "")
        if n.startswith("k_decode"): print(n, "calls", r["Calls"], "avg ms", round(float(r["AverageNs"]) / 1e6, 4))
PY
