set -u
mkdir -p gpurun_out/r02m
for n in 3072 4096 7168 8192; do for k in product lib_lanestride3.so lib_lanestride17.so; do
  if [ "$k" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/$k; fi
  MI_RTJ_LIB=$L timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --frames $n --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('frames=$n $k', 'fps', d['value'], d['speculative_index']['packets_proven'], 'per4096:', {a:round(b['ms']/$n*4096,4) for a,b in k.items() if b['ms']>0.05})"
done; done 2>&1 | tee gpurun_out/r02m/ab_lane_stride.txt
