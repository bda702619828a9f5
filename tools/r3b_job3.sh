#!/bin/bash
# is the copy out of a session slower than back-to-back copies because of where the host memory / the thread is?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
{
lscpu | grep -i -E "numa|socket|model name|^CPU\(s\)"
which numactl taskset
for d in /sys/bus/pci/devices/*; do v=$(cat $d/vendor 2>/dev/null); c=$(cat $d/class 2>/dev/null); if [ "$v" = "0x1002" ] && { [ "${c:0:6}" = "0x0302" ] || [ "${c:0:6}" = "0x0380" ] || [ "${c:0:6}" = "0x1200" ]; }; then echo "$d numa_node=$(cat $d/numa_node) local_cpulist=$(cat $d/local_cpulist)"; fi; done
for n in /sys/devices/system/node/node*; do echo "$n $(cat $n/cpulist)"; done
nproc; cat /proc/self/status | grep -i allowed_list
} > $O/numa.txt 2>&1
cat $O/numa.txt
run() { # label, env...
  local label=$1; shift
  env "$@" python - "$label" <<'PY' | tee -a $O/e2e_numa.txt
import sys, json, os
sys.path.insert(0, '.')
aff = os.environ.get("AFF")
if aff:
    cpus = set()
    for part in aff.split(","):
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    os.sched_setaffinity(0, cpus)
import tools.e2e_bench as E
r = E.run(int(os.environ.get("W", "1920")), int(os.environ.get("H", "1088")), packets=int(os.environ.get("PK", "64")), repeat=int(os.environ.get("REP", "32")), depth=int(os.environ.get("DEPTH", "6")), flavours=("_pipe",), two_streams=False)
one = [v for k, v in r.items() if isinstance(v, dict) and 'in flight' in k][0]
print(f"{sys.argv[1]:60s} {one.get('fps')}")
PY
}
GPU_CPUS=$(for d in /sys/bus/pci/devices/*; do v=$(cat $d/vendor 2>/dev/null); c=$(cat $d/class 2>/dev/null); if [ "$v" = "0x1002" ] && [ "${c:0:6}" = "0x0302" -o "${c:0:6}" = "0x0380" ]; then cat $d/local_cpulist; fi; done | head -1)
echo "GPU local cpus: $GPU_CPUS" | tee -a $O/e2e_numa.txt
K4="W=3840 H=2160 PK=24 REP=8"
run "4K skip=6 singles (as is)" $K4 MI_RTJ_EXP_SKIP=6 MI_RTJ_OUT_GROUP=1
run "4K skip=6 singles one out stream" $K4 MI_RTJ_EXP_SKIP=6 MI_RTJ_OUT_GROUP=1 MI_RTJ_OUT_STREAMS=1
[ -n "$GPU_CPUS" ] && run "4K skip=6 singles on the GPU's cpus" $K4 MI_RTJ_EXP_SKIP=6 MI_RTJ_OUT_GROUP=1 AFF=$GPU_CPUS
for n in /sys/devices/system/node/node*; do
  run "4K skip=6 singles on $(basename $n)" $K4 MI_RTJ_EXP_SKIP=6 MI_RTJ_OUT_GROUP=1 AFF=$(cat $n/cpulist)
done
[ -n "$GPU_CPUS" ] && run "4K full pairs on the GPU's cpus" $K4 AFF=$GPU_CPUS
[ -n "$GPU_CPUS" ] && run "1080p full pairs on the GPU's cpus" AFF=$GPU_CPUS
run "1080p full pairs (as is)" X=1
