#!/bin/bash
# extra SQ counters (stall reasons) — separate passes, no tracing
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcx
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu"
run() { local name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_THREAD_CYCLES_VALU
run b SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU
run c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES
python tools/pmc_summary.py $OUT gpurun_out/pmcx_summary.json > /dev/null
