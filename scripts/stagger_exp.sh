#!/bin/bash
# staggered chains: evals/s by batch with the measured stagger (default), without (0) and chains off
run() { NSG_CHAINS=$1 NSG_CHAIN_DELAY_US=$2 python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --batch $3 --precision ${4:-f16m8} --steps 30 --warmup 10 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('prec=${4:-f16m8} chains=$1 delay=$2 batch=$3', round(d['value']), flush=True)"; }
for rep in 1 2; do
for b in 512 640 768 1024; do
run 2 -1 $b; run 2 0 $b; run 1 0 $b
done
done
for b in 512 1024; do run 2 -1 $b f16x3; run 2 0 $b f16x3; run 1 0 $b f16x3; run 2 -1 $b bf16; run 1 0 $b bf16; run 2 -1 $b fp32; run 1 0 $b fp32; done
