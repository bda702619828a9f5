#!/bin/bash
# round 4, last: smoke, the whole GPU suite, the
# profile round with the round's final sources, the DV line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_all.log
bash tools/profile_round.sh r04
timeout -k 10 300 python bench.py --config dv > gpurun_out/r04/bench_dv.json 2> gpurun_out/r04/bench_dv.err; echo "dv rc=$?"
