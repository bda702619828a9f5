#!/bin/bash
# round 4: what bounds k_decode_split?  write ceiling, the two roles alone, stall counters
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
python - <<'PY' > $O/fill_bw.txt 2>&1
import torch, time
x = torch.empty(48*1024**3, dtype=torch.uint8, device="cuda")
for i in range(2): x.fill_(i)
torch.cuda.synchronize()
t=time.time()
for i in range(5): x.fill_(i+3)
torch.cuda.synchronize()
dt=(time.time()-t)/5
print("fill", x.numel()/dt/1e12, "TB/s", dt*1e3, "ms")
y = torch.empty(24*1024**3, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t=time.time()
for i in range(5): y.copy_(x[:y.numel()])
torch.cuda.synchronize(); dt=(time.time()-t)/5
print("copy r+w", 2*y.numel()/dt/1e12, "TB/s")
PY
cat $O/fill_bw.txt
B="--no-cpu --no-stress --no-e2e --no-sweep --steps 6 --warmup 2"
L=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_exp.so
for only in 1 2; do
  MI_RTJ_LIB=$L MI_RTJ_SPLIT_ONLY=$only timeout -k 10 200 python bench.py $B > $O/only$only.json 2> $O/only$only.err
  echo "only=$only rc=$?"; tail -c 600 $O/only$only.json; tail -3 $O/only$only.err
done
rocprofv3-avail list > $O/avail.txt 2>&1 || rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -c . $O/avail.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  d=$O/pmc3_$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 2 --warmup 1 > $d.log 2>&1 || echo "pmc set failed: $set"
done
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/r4/pmc3_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_decode_split" not in k and "k_spec_walk" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k in acc:
    print(k, {c: round(v/max(n[k][c],1)) for c,v in acc[k].items()})
PY
