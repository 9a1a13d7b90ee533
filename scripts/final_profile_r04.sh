#!/bin/bash
# Round-4 measurement set for profiles/r04 (run on the GPU box via gpurun): scripts/final_profile_r04.sh
# -> gpurun_out/final_r04/: rocprofv3 --kernel-trace --stats of the benchmark, PMC passes over bench.py's own workload
#    (separate runs, no trace domains), the same for BASELINE configs[4] (40x384 bf16, batch 1024) and configs[1],
#    batch curve
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/final_r04
mkdir -p $OUT
kt() { # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$name -- python3 bench.py "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/kt_$name.err
  cp $(find $OUT/kt_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv; echo "kernel trace $name: $?"
  rm -rf $OUT/kt_$name
}
kt f16m6_b512_20x256_40steps --selfplay-seconds 0 --no-cpu-baseline --no-host-path --no-clock-sample --no-other-configs --steps 40 --warmup 10
kt bf16_b1024_40x384 --workload-only 12 --net 40x384 --batch 1024 --precision bf16
kt f16m6_b64_10x192 --workload-only 40 --net 10x192 --batch 64 --precision f16m6
kt f16m6_b128_20x256 --workload-only 40 --net 20x256 --batch 128 --precision f16m6
scripts/pmc.sh r04_f16m6 512 f16m6 > $OUT/pmc.log 2>&1 && python3 scripts/pmc_summary.py gpurun_out/pmc_r04_f16m6 > $OUT/pmc_f16m6_conv_top.txt && cp gpurun_out/pmc_r04_f16m6/summary.json $OUT/pmc_f16m6_summary.json; echo "pmc: $?"
PMC_PROGRAM="bench.py --workload-only 4 --net 40x384 --batch 1024 --precision bf16" scripts/pmc.sh r04_bf16_40x384 1024 bf16 > $OUT/pmc_bf16.log 2>&1 && python3 scripts/pmc_summary.py gpurun_out/pmc_r04_bf16_40x384 > $OUT/pmc_bf16_40x384_b1024_top.txt && cp gpurun_out/pmc_r04_bf16_40x384/summary.json $OUT/pmc_bf16_40x384_b1024_summary.json; echo "pmc bf16: $?"
PMC_PROGRAM="bench.py --workload-only 6 --net 20x256 --batch 128 --precision f16m6" scripts/pmc.sh r04_f16m6_b128 128 f16m6 > $OUT/pmc_b128.log 2>&1 && python3 scripts/pmc_summary.py gpurun_out/pmc_r04_f16m6_b128 > $OUT/pmc_f16m6_b128_top.txt && cp gpurun_out/pmc_r04_f16m6_b128/summary.json $OUT/pmc_f16m6_b128_summary.json; echo "pmc b128: $?"
python3 scripts/batch_sweep.py f16m6 --batches 1,2,4,8,16,17,24,32,64,65,96,128,129,144,160,192,256,257,320,384,512,576,640,768,1024 --rounds 3 > $OUT/batch_curve_f16m6.txt 2>&1; echo "sweep: $?"
ls $OUT
