#!/bin/bash
# the round's profile set with the final kernels + randomised differential runs of the speculative index
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh r03 2>&1 | tail -3
O=gpurun_out/r03
for mode in 2 1 3 4; do timeout -k 10 130 python tools/stress_spec.py 100 $((100 + mode)) $mode 2>&1 | tail -1 | sed "s/^/MI_RTJ_SPEC=$mode: /" | tee -a $O/stress_spec_runs.txt; done
timeout -k 10 200 python bench.py --config streams4k > $O/bench_streams4k.json 2>$O/bench_streams4k.err; echo "streams4k rc=$?"
timeout -k 10 200 python bench.py --config mixed > $O/bench_mixed.json 2>$O/bench_mixed.err; echo "mixed rc=$?"
bash tools/noisy_content.sh > $O/noisy_content.txt 2>&1; echo "noisy rc=$?"; tail -12 $O/noisy_content.txt
