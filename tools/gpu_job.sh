set -u
mkdir -p gpurun_out/r02k
for d in 4 6 8; do timeout -k 10 300 python tools/e2e_bench.py --depth $d; done 2>&1 | tee gpurun_out/r02k/e2e3.txt
(timeout -k 10 900 python -m pytest tests/test_plugin_harness.py -m gpu -x -q 2>&1 | tail -3)
