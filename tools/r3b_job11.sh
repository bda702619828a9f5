#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
one() { python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'])" | tee -a $O/streams4k_var.txt; }
for d in /sys/bus/pci/devices/*; do v=$(cat $d/vendor 2>/dev/null); c=$(cat $d/class 2>/dev/null); [ "$v" = "0x1002" ] && echo "$d class=$c numa=$(cat $d/numa_node)"; done | grep -v "class=0x06" | tee -a $O/streams4k_var.txt
rocm-smi --showbus 2>/dev/null | grep -i "GPU\[" | tee -a $O/streams4k_var.txt
python bench.py --config streams4k --no-cpu --steps 6 --warmup 2 2>/dev/null | one "as is, 6 steps"
python bench.py --config streams4k --no-cpu --steps 20 --warmup 3 2>/dev/null | one "as is, 20 steps"
taskset -c 0-63,128-191 python bench.py --config streams4k --no-cpu --steps 10 --warmup 2 2>/dev/null | one "node0 cpus"
taskset -c 64-127,192-255 python bench.py --config streams4k --no-cpu --steps 10 --warmup 2 2>/dev/null | one "node1 cpus"
MI_RTJ_DEPTH_OVERRIDE=8 python bench.py --config streams4k --no-cpu --steps 10 --warmup 2 2>/dev/null | one "as is depth 8"
