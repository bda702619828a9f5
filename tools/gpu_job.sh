set -u
mkdir -p gpurun_out/r02u
(timeout -k 10 600 python -m pytest tests/test_gpu_variant_paths.py tests/test_gpu_parity.py tests/test_plugin_harness.py -m gpu -x -q > gpurun_out/r02u/pytest_quick.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02u/pytest_quick.log; tail -5 gpurun_out/r02u/pytest_quick.log)
bash tools/ab_libs.sh 2 -- lib_head.so product 2>&1 | tee gpurun_out/r02u/ab.txt
