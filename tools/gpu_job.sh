set -u
mkdir -p gpurun_out/r02f
bash tools/ab_libs.sh 2 --no-stress --no-e2e -- product lib_ldspad900.so lib_ldspad2600.so 2>&1 | tee gpurun_out/r02f/ab_occupancy.txt
