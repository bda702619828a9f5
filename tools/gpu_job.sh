set -u
mkdir -p gpurun_out/r02j
tools/ubench/idct_asm_rate | head -3 | tee gpurun_out/r02j/idct_asm_rate.txt
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02j/pytest.log; tail -4 gpurun_out/r02j/pytest.log)
bash tools/ab_libs.sh 2 -- lib_asm_off.so product lib_px1.so 2>&1 | tee gpurun_out/r02j/ab_px.txt
