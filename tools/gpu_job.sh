set -u
rm -rf gpurun_out/pmc gpurun_out/r02
bash tools/profile_round.sh r02
