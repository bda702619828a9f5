set -u
mkdir -p gpurun_out/r02t
MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_stamps.so timeout -k 10 300 python tools/stamps.py 8 2>&1 | tee gpurun_out/r02t/stamps_amp8.txt
