#!/bin/bash
# Collects PMC counters for one bench configuration (run on the GPU box via gpurun).
# usage: scripts/pmc.sh <tag> <bench args...>     -> gpurun_out/pmc_<tag>/
# Counters are collected in their own runs (no trace domains), one --pmc pass each.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path "${BARGS[@]}" > $OUT/$name.log 2>&1
}
BARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
find $OUT -name "*counter_collection.csv" | head
