#!/bin/bash
# round 4, last: the randomised differential run (index and pictures against the CPU decoder, default and forced split / serial modes), then the whole GPU suite once more with the final library
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 400 python tools/stress_spec.py 300 41 > $O/stress_default.log 2>&1; echo "stress default rc=$?"; tail -2 $O/stress_default.log
MI_RTJ_ROTATE=1 MI_RTJ_SPLIT=1 timeout -k 10 400 python tools/stress_spec.py 200 42 > $O/stress_split.log 2>&1; echo "stress forced split form rc=$?"; tail -2 $O/stress_split.log
MI_RTJ_INDEX=serial timeout -k 10 400 python tools/stress_spec.py 200 43 > $O/stress_serial.log 2>&1; echo "stress serial walker rc=$?"; tail -2 $O/stress_serial.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_last.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_last.log
