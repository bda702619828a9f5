set -u
mkdir -p gpurun_out/r02q2
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02q2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02q2/pytest.log; tail -3 gpurun_out/r02q2/pytest.log)
timeout -k 10 400 python tools/stress_spec.py 200 31 1 2>&1 | tail -2 | tee gpurun_out/r02q2/stress1.txt
timeout -k 10 400 python tools/stress_spec.py 200 32 3 2>&1 | tail -2 | tee gpurun_out/r02q2/stress3.txt
