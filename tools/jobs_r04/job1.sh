#!/bin/bash
# round 4, first GPU job: parity of the split decode (luma waves + pooling chroma waves), then the bench A/B
set -o pipefail
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dc_only.py tests/test_gpu_overlap.py tests/test_gpu_spec_index.py tests/test_gpu_variant_paths.py -m gpu -x -q > gpurun_out/r4/pytest1.log 2>&1
rc=$?
tail -5 gpurun_out/r4/pytest1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-stress --no-e2e --no-sweep --steps 10 --warmup 3 > gpurun_out/r4/bench_split.json 2> gpurun_out/r4/bench_split.err && \
MI_RTJ_SPLIT=0 timeout -k 10 300 python bench.py --no-stress --no-e2e --no-sweep --no-cpu --steps 10 --warmup 3 > gpurun_out/r4/bench_classic.json 2> gpurun_out/r4/bench_classic.err
rc=$?
python - <<'PY'
import json
for n in ("split","classic"):
    try:
        j=json.loads(open(f"gpurun_out/r4/bench_{n}.json").read().strip().split("\n")[-1])
        print(n, j["value"], j.get("kernels_ms"), j.get("roofline"))
    except Exception as e:
        print(n, "unreadable", e)
PY
exit $rc
