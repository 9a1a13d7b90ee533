#!/bin/bash
# staggered chains below a full chip and on the other nets: default rule vs one chain
run() { NSG_CHAINS=$1 python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --batch $2 --precision ${3:-f16m8} --net ${4:-20}x${5:-256} --steps 30 --warmup 10 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('net=${4:-20}x${5:-256} prec=${3:-f16m8} chains<=$1 batch=$2', round(d['value']), flush=True)"; }
for b in 272 320 384 448 512; do run 2 $b; run 1 $b; done
for b in 384 512; do run 2 $b f16x3; run 1 $b f16x3; done
run 2 512 f16m8 10 192; run 1 512 f16m8 10 192
run 2 256 bf16 40 384; run 1 256 bf16 40 384; run 2 256 f16m8 40 384; run 1 256 f16m8 40 384
