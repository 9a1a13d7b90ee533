#!/bin/bash
# the BASELINE configs other than the headline: 10x192 at batch 64, 40x384 at batch 1024 (bf16 and the split formats)
run() { python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --sustain-seconds 0 --net $1 --batch $2 --precision $3 --steps ${4:-30} --warmup 5 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('net=$1 batch=$2 prec=$3', round(d['value']), 'evals/s  conv', round(r['avg_launch_ms'],4), 'ms  frac', round(r['frac'],3), ' whole-net TF', round(d['whole_net_tflops'],1), flush=True)"; }
run 10x192 64 f16m6 60; run 10x192 512 f16m6; run 10x192 64 f16x3 60
run 40x384 1024 bf16 10; run 40x384 1024 f16m6 10; run 40x384 1024 f16m8 10; run 40x384 1024 f16x3 10
run 20x256 256 f16m6; run 20x256 640 f16m6; run 20x256 1024 f16m6
