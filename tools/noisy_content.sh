#!/bin/bash
# Noisier synthetic content through the same bench (one JSON line each): how the speculative index holds up
# as blocks stop ending in zero runs.  bash tools/noisy_content.sh > out.jsonl
set -u
cd "$(dirname "$0")/.."
for amp in 12 16 20 24 32; do
  timeout -k 10 300 python bench.py --no-cpu --amp $amp 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({\"amp\": $amp, \"fps\": d[\"value\"], \"avg_packet\": d[\"config\"][\"avg_packet_bytes\"], \"spec\": d[\"speculative_index\"], \"kernels_ms\": {a:b[\"ms\"] for a,b in d[\"kernels\"].items()}}))"
done
