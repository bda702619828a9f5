set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02c
for v in product NO_PARSE NO_TRANSFORM; do
  if [ "$v" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_$v.so; fi
  export MI_RTJ_LIB=$L
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r02c/$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-stress --no-e2e > gpurun_out/r02c/$v.log 2>&1
  echo "$v rc=$?"
done
python - <<'PY'
import csv, glob
for v in ('product','NO_PARSE','NO_TRANSFORM'):
    acc={}
    for f in glob.glob(f'gpurun_out/r02c/{v}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_decode' in r['Kernel_Name'] and 'list' not in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    print(v, {k: round(sum(x)/len(x)/1e6,1) for k,x in acc.items()}, 'dispatches', {k:len(x) for k,x in acc.items()})
PY
