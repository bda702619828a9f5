#!/bin/bash
# The path on noisier content (Q=255; noise amplitude of the synthetic frames), 1024 pictures per launch, after the
# plan's policy has settled (8 warm-up launches):  bash tools/noisy_content.sh > gpurun_out/<tag>/noisy_content.txt
cd "$(dirname "$0")/.."
for amp in 8 12 16 20 24 32 40 48 64; do
python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 5 --warmup 8 --frames 1024 --amp $amp 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['speculative_index']
print('amp %2d' % $amp, '%8.0f frames/s' % d['value'], 'packet %7d B' % d['config']['avg_packet_bytes'], 'proven', s['packets_proven'], 'repaired', s['chunks_repaired'], 'lead', s['walker_lead_bytes'], {a:b['ms'] for a,b in d['kernels'].items() if b['ms']>0.02})"
done
