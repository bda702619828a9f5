#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputests_final.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/gputests_final.log; tail -4 gpurun_out/r03/gputests_final.log
grep -q "rc=0" gpurun_out/r03/gputests_final.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r03 2>&1 | grep -v "^{" | tail -3
