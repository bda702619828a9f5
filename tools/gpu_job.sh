set -u
mkdir -p gpurun_out/r02y
(timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02y/pytest_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02y/pytest_parity.log; tail -5 gpurun_out/r02y/pytest_parity.log)
