set -u
mkdir -p gpurun_out/r02a
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log; tail -3 gpurun_out/r02a/pytest.log)
bash tools/ab_libs.sh 2 -- lib_r1base.so product > gpurun_out/r02a/ab.txt 2>&1
cat gpurun_out/r02a/ab.txt
tools/ubench/valu_kinds2 4 > gpurun_out/r02a/valu_kinds2_4w.txt 2>&1
tools/ubench/valu_kinds2 8 > gpurun_out/r02a/valu_kinds2_8w.txt 2>&1
tools/ubench/valu_kinds2 2 > gpurun_out/r02a/valu_kinds2_2w.txt 2>&1
echo done
