#!/bin/bash
# Other shapes / contents through the same bench (one JSON line each): bash tools/other_configs.sh > out.jsonl
set -u
cd "$(dirname "$0")/.."
for a in "--width 3840 --height 2160 --frames 4096" "--width 3840 --height 2160 --frames 64" "--quality 128" "--quality 64" "--width 320 --height 240 --frames 65535" "--width 320 --height 240 --frames 4096" "--amp 0" "--width 1280 --height 720"; do
  timeout -k 10 300 python bench.py --no-cpu --no-stress --no-e2e --no-sweep --steps 10 $a 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({\"args\": \"$a\", \"fps\": d[\"value\"], \"mpix_s\": d[\"mpixels_per_s\"], \"avg_packet\": d[\"config\"][\"avg_packet_bytes\"], \"path_gbs\": d[\"path_gbs\"], \"kernels_ms\": {a:b[\"ms\"] for a,b in d[\"kernels\"].items()}}))"
done
