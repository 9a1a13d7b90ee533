#!/bin/bash
# Collects PMC counters for the benchmark's workload (run on the GPU box via gpurun).
# usage: scripts/pmc.sh <tag> [batch] [precision]     -> gpurun_out/pmc_<tag>/ (+ summary.json by scripts/pmc_summary.py)
# Counters are collected in their own runs (no trace domains), one --pmc pass each.  The profiled program is
# bench.py itself in its --workload-only form: the bench line's workload (20x256, synthetic weights of seed 0, the
# initial position in every slot, device-resident forwards), six steps, without the legs that start child processes or
# bracket launches with HIP events (PMC_PROGRAM="scripts/one_forward.py 512 f16m6 6" profiles round 3's stand-in).
set -euo pipefail
TAG=$1; B=${2:-512}; PREC=${3:-f16m6}
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 ${PMC_PROGRAM:-bench.py --workload-only 6 --batch $B --precision $PREC} > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_summary.py $OUT
