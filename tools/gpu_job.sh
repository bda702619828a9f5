set -u
mkdir -p gpurun_out/r02w
(timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02w/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02w/pytest.log; tail -4 gpurun_out/r02w/pytest.log)
rm -rf gpurun_out/pmc gpurun_out/r02
bash tools/profile_round.sh r02
