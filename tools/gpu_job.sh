set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02l
for v in product flush16; do
  if [ "$v" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_$v.so; fi
  export MI_RTJ_LIB=$L
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r02l/$v-$c -- python3 bench.py --frames 4096 --steps 3 --warmup 1 --no-cpu --no-stress --no-e2e > gpurun_out/r02l/$v-$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob
for v in ('product','flush16'):
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        acc=[]
        for f in glob.glob(f'gpurun_out/r02l/{v}-{c}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if 'k_spec_walk' in r['Kernel_Name'] and '768' in r['Kernel_Name'] and r['Counter_Name']==c:
                    acc.append(float(r['Counter_Value']))
        print(v, c, round(sum(acc)/max(len(acc),1)/1e6,3), 'GiB-ish (KiB/1e6)', len(acc))
PY
unset MI_RTJ_LIB
for rep in 1 2 3; do for v in product flush16; do
  if [ "$v" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_$v.so; fi
  MI_RTJ_LIB=$L timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --steps 6 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', 'fps', d['value'], {a:round(b['ms']/2,4) for a,b in k.items() if b['ms']>0.05})"
done; done 2>&1 | tee gpurun_out/r02l/ab_flush16.txt
