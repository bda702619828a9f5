#!/bin/bash
# round 4: other shapes and qualities, split form against the classic one (the luma / chroma wave counts differ by shape)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
: > $O/other_shapes_ab.txt
for a in "--width 3840 --height 2160 --frames 4096" "--width 1280 --height 720" "--width 720 --height 576 --frames 32768" "--width 320 --height 240 --frames 65536" "--width 4096 --height 2176 --frames 2048" "--quality 128" "--amp 0 --content hash"; do
for sp in 1 0; do
  MI_RTJ_SPLIT=$sp timeout -k 10 300 python bench.py --no-stress --no-e2e --no-sweep --steps 10 --verify-frames 64 --cpu-seconds 1 $a 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(json.dumps({'args': '$a', 'split': $sp, 'fps': d['value'], 'k_decode_ms': d['kernels']['k_decode']['ms'], 'frac': d['roofline']['frac'], 'mismatches': d.get('parity_mismatches'), 'checked': d.get('parity_checked')}))" | tee -a $O/other_shapes_ab.txt
done; done
