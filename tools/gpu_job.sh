set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02r2
MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_flushtog.so timeout -k 10 300 python -m pytest tests/test_gpu_spec_index.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do for k in product lib_flushtog.so; do
  if [ "$k" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/$k; fi
  MI_RTJ_LIB=$L timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$k', 'fps', d['value'], d['speculative_index']['packets_proven'], 'per4096:', {a:round(b['ms']/4,4) for a,b in k.items() if b['ms']>0.05})"
done; done 2>&1 | tee gpurun_out/r02r2/ab_flush_together.txt
for v in product flushtog; do
  if [ "$v" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_$v.so; fi
  export MI_RTJ_LIB=$L
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r02r2/$v-$c -- python3 bench.py --frames 4096 --steps 3 --warmup 1 --no-cpu --no-stress --no-e2e > gpurun_out/r02r2/$v-$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob
for v in ('product','flushtog'):
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        acc=[]
        for f in glob.glob(f'gpurun_out/r02r2/{v}-{c}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if 'k_spec_walk' in r['Kernel_Name'] and '768' in r['Kernel_Name'] and r['Counter_Name']==c:
                    acc.append(float(r['Counter_Value']))
        print(v, c, round(sum(acc)/max(len(acc),1)/1e6,3), len(acc))
PY
