#!/bin/bash
# Round-3 measurement set for profiles/r03 (run on the GPU box via gpurun): scripts/final_profile_r03.sh
# -> gpurun_out/final_r03/: default bench line, rocprofv3 --kernel-trace --stats of the benchmark, PMC passes
#    (separate runs, no trace domains), PMC of the extract kernel, stamps, batch curve
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/final_r03
mkdir -p $OUT
python3 bench.py > $OUT/f_bench_default_f16m6.json 2> $OUT/bench_default.err; echo "bench default: $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --no-clock-sample --steps 40 --warmup 10 > $OUT/j_bench_f16m6_under_rocprof_40steps.json 2> $OUT/kt.err
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/j_f16m6_b512_20x256_kernel_stats_40steps.csv; echo "kernel trace: $?"
rm -rf $OUT/kt
scripts/pmc.sh r03_f16m6 --selfplay-seconds 0 > $OUT/pmc.log 2>&1 && python3 scripts/pmc_summary.py gpurun_out/pmc_r03_f16m6 > $OUT/pmc_f16m6_conv_top.txt && cp gpurun_out/pmc_r03_f16m6/summary.json $OUT/pmc_f16m6_summary.json; echo "pmc: $?"
scripts/pmc_extract.sh > $OUT/pmc_extract_top.txt 2>&1 && cp gpurun_out/pmc_extract/summary.json $OUT/pmc_extract_summary.json; echo "pmc extract: $?"
for b in 1 128 512; do echo "== B=$b"; python3 scripts/stamps.py --precision f16m6 --batch $b 2>&1 | tail -1; done > $OUT/i_stamps_f16m6_b1_b128_b512.txt
python3 scripts/batch_sweep.py f16m6 --batches 1,8,16,32,33,64,65,96,128,129,160,192,256,257,320,384,512,576,640,768,1024 --rounds 3 > $OUT/o_batch_curve_f16m6.txt 2>&1; echo "sweep: $?"
ls $OUT
