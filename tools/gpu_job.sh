set -u
mkdir -p gpurun_out/r02m
for rep in 1 2; do for wg in 0 1; do echo "MI_RTJ_DEC_WG=$wg"; MI_RTJ_DEC_WG=$wg bash tools/ab_libs.sh 1 -- product; done; MI_RTJ_DEC_WG=1 bash tools/ab_libs.sh 1 -- lib_wg4.so lib_wg8.so; done 2>&1 | tee gpurun_out/r02m/ab_wg.txt
