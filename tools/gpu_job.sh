set -u
mkdir -p gpurun_out/r02p
python tools/pcie_probe.py 2>/dev/null | tee gpurun_out/r02p/pcie_probe.json
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02p/pytest.log; tail -8 gpurun_out/r02p/pytest.log)
bash tools/ab_libs.sh 2 -- lib_sparse0.so lib_sparse16.so product lib_sparse32.so 2>&1 | tee gpurun_out/r02p/ab_sparse.txt
