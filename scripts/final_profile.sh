#!/bin/bash
# One-shot measurement set for profiles/: scripts/final_profile.sh <tag>   (run on the GPU box via gpurun)
# -> gpurun_out/final_<tag>/ : bench JSON lines, rocprofv3 --kernel-trace --stats summaries, PMC passes, stamps
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench_default_f16m8.json 2> $OUT/bench_default.err
echo "bench default done"
python3 bench.py --precision f16x3 --selfplay-seconds 0 --no-cpu-baseline > $OUT/bench_f16x3.json 2>> $OUT/bench_default.err
python3 bench.py --precision fp32 --selfplay-seconds 0 --no-cpu-baseline > $OUT/bench_fp32.json 2>> $OUT/bench_default.err
for prec in f16m8 f16x3 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$prec -- python3 bench.py --precision $prec --selfplay-seconds 0 --no-cpu-baseline --no-host-path --steps 6 --warmup 2 > $OUT/kt_$prec.log 2>&1
  cp $(find $OUT/kt_$prec -name "*kernel_stats.csv" | head -1) $OUT/${prec}_b512_20x256_kernel_stats.csv
  echo "kernel trace $prec done"
done
for prec in f16m8 f16x3; do
scripts/pmc.sh ${TAG}_$prec --precision $prec --selfplay-seconds 0 > /dev/null 2>&1 && python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_$prec > $OUT/pmc_${prec}_conv_top.txt && cp gpurun_out/pmc_${TAG}_$prec/summary.json $OUT/pmc_${prec}_summary.json
echo "pmc $prec done"
done
scripts/pmc.sh ${TAG}_fp32 --precision fp32 --selfplay-seconds 0 > /dev/null 2>&1 && python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_fp32 > $OUT/pmc_fp32_conv_top.txt && cp gpurun_out/pmc_${TAG}_fp32/summary.json $OUT/pmc_fp32_summary.json
echo "pmc fp32 done"
python3 scripts/stamps.py --precision f16m8 > $OUT/stamps_f16m8_b512.txt 2>&1
python3 scripts/stamps.py --precision f16x3 > $OUT/stamps_f16x3_b512.txt 2>&1
python3 scripts/stamps.py --batch 64 --net 10x192 > $OUT/stamps_f16x3_10x192_b64.txt 2>&1
rm -rf $OUT/kt_f16x3 $OUT/kt_fp32
ls $OUT
