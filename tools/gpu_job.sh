set -u
bash tools/profile_round.sh r02 2>&1 | tail -5
python tools/make_traffic.py gpurun_out/r02/pmc_summary.json gpurun_out/r02/traffic.json "r02 (round 2 final kernels)" 4096
ls gpurun_out/r02
