set -u
mkdir -p gpurun_out/r02l
(timeout -k 10 1000 python -m pytest tests/test_bench_launcher.py -m gpu -x -q > gpurun_out/r02l/pytest_bench.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02l/pytest_bench.log; tail -8 gpurun_out/r02l/pytest_bench.log)
timeout -k 10 300 python bench.py --config streams4k 2>/dev/null | tee gpurun_out/r02l/bench_streams4k.json
timeout -k 10 300 python bench.py --config mixed 2>/dev/null | tee gpurun_out/r02l/bench_mixed.json
