#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "enc or config or stream" > $O/pytest_enc.log 2>&1; echo "encoder tests rc=$?"; tail -5 $O/pytest_enc.log
timeout -k 10 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"; tail -3 $O/bench_full.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4/bench_full.json").read().strip().split("\n")[-1])
for k in ("value", "ms_per_step", "roofline", "encoder", "stress_amp64", "cpu_baseline", "parity_checked", "parity_mismatches", "by_batch", "end_to_end", "copy_ceiling_gbs", "kernels"):
    print(k, json.dumps(j.get(k))[:900])
PY
