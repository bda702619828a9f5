#!/bin/bash
# PMC passes of k_decode with only the chroma / only the luma parts running.  The two libraries are built by
# tools/ab_variants.sh with V=( [chroma]="-DMIRTJ_EXP_ONLY_PART=2" [luma]="-DMIRTJ_EXP_ONLY_PART=0" ) after adding,
# behind k_decode's "if (slot >= ngroups) return;", the three lines
#   #ifdef MIRTJ_EXP_ONLY_PART
#     if ((MIRTJ_EXP_ONLY_PART == 2) != (part == 2u)) return;
#   #endif
# (not kept in the source: the pictures such a build makes are incomplete).  Result: profiles/r01/v20_pmc_luma_vs_chroma.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_parts
rm -rf $OUT; mkdir -p $OUT
for v in chroma luma; do
  export MI_RTJ_LIB=$GRAFT_REPO_ROOT/gmerlin-avdecoder_amd/lib/ab/lib_$v.so
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/$v/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/$v.sq1.log 2>&1; echo "$v sq1 rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$v/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/$v.sq2.log 2>&1; echo "$v sq2 rc=$?"
  timeout -k 10 200 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/$v/sq3 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/$v.sq3.log 2>&1; echo "$v sq3 rc=$?"
  python tools/pmc_summary.py $OUT/$v $OUT/$v.json > /dev/null
done
python - <<'PY'
import json
for v in ("chroma", "luma"):
    d = json.load(open(f"gpurun_out/pmc_parts/{v}.json"))["k_decode"]
    print(v, {k: round(x) for k, x in d.items()})
PY
