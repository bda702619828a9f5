set -u
mkdir -p gpurun_out/r02z
bash tools/ab_libs.sh 2 --no-stress --no-e2e -- product lib_stagger.so 2>&1 | tee gpurun_out/r02z/ab_stagger.txt
