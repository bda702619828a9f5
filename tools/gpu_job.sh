set -u
mkdir -p gpurun_out/r02g
timeout -k 10 500 python tools/stress_spec.py 400 21 3 2>&1 | tail -3 | tee gpurun_out/r02g/stress_long_lead.txt
MI_RTJ_ROTATE=1 MI_RTJ_DEFER=0 timeout -k 10 500 python tools/stress_spec.py 400 22 1 2>&1 | tail -3 | tee gpurun_out/r02g/stress_rotate_short_lead.txt
