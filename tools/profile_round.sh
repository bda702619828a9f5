#!/bin/bash
# One GPU-box pass that produces everything committed under profiles/<tag>/:
#   bench.json (default bench.py run, with cpu_baseline), kernel_stats.csv (rocprofv3 --kernel-trace --stats
#   of the same command), pmc_summary.json (separate --pmc passes, tools/pmc_passes.sh).
# Usage on the GPU box: bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu --no-stress --no-e2e --no-sweep > $OUT/trace.log 2>&1; echo "trace rc=$?"
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
bash tools/pmc_passes.sh --steps 3 --warmup 1 --no-cpu --no-stress --no-e2e --no-sweep > $OUT/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc $OUT/pmc_summary.json > /dev/null
tail -1 $OUT/bench.json
