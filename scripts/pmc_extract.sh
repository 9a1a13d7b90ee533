#!/bin/bash
# HBM bytes of the plane-expansion kernel by PMC (separate passes, no trace domains): scripts/pmc_extract.sh
# -> gpurun_out/pmc_extract/summary.json  (copy to profiles/rNN/pmc_extract_summary.json)
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_extract
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/extract_pmc.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scripts/extract_pmc.py > $OUT/write.log 2>&1
python3 scripts/pmc_summary.py $OUT
