#!/bin/bash
# round 4: the new policy test
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_decode_policy.py -m gpu -x -q 2>&1 | tail -15
