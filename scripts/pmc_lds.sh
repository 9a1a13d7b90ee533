#!/bin/bash
# LDS bank-conflict counters only: scripts/pmc_lds.sh <tag> <lib> <bench args...>
set -e
TAG=$1; LIB=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NSG_LIB=$LIB
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --selfplay-seconds 0 "$@" > $OUT/sq2.log 2>&1
python3 scripts/pmc_summary.py $OUT | head -24
