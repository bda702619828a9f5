set -u
mkdir -p gpurun_out/r02s2
MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_verb8.so timeout -k 10 300 python -m pytest tests/test_gpu_spec_index.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do for k in product lib_verb4.so lib_verb8.so; do
  if [ "$k" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/$k; fi
  MI_RTJ_LIB=$L timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$k', 'fps', d['value'], d['speculative_index']['packets_proven'], 'per4096:', {a:round(b['ms']/4,4) for a,b in k.items() if b['ms']>0.05})"
done; done 2>&1 | tee gpurun_out/r02s2/ab_verify_batch.txt
