#!/bin/bash
# A/B of libraries on noisy content: bash tools/noisy_ab.sh lib_a.so lib_b.so ... (relative to lib/ab/; "product")
cd "$(dirname "$0")/.."
for k in "$@"; do
if [ "$k" = product ]; then L=$PWD/gmerlin-avdecoder_amd/lib/libmi_rtjpeg.so; else L=$PWD/gmerlin-avdecoder_amd/lib/ab/$k; fi
for amp in 28 32 36 40; do
MI_RTJ_LIB=$L python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 5 --warmup 8 --frames 1024 --amp $amp 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['speculative_index']
print('$k amp %2d' % $amp, '%8.0f frames/s' % d['value'], 'proven', s['packets_proven'], 'repaired', s['chunks_repaired'], 'lead', s['walker_lead_bytes'])"
done; done
