#!/bin/bash
# persistent trunk kernel (one launch for all 3x3 layers) vs per-layer launches, with start skew
run() { NSG_TRUNK_KERNEL=$1 NSG_TRUNK_SKEW_US=$2 python bench.py --precision ${3:-f16m8} --selfplay-seconds 0 --no-cpu-baseline --no-host-path --steps 20 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('trunk_kernel=$1 skew_us=$2 ${3:-f16m8}', round(d['value']), round(d['ms_per_step'],3), flush=True)"; }
for rep in 1 2; do
run 0 0; run 1 0; run 1 4; run 1 8; run 1 12; run 1 20
done
run 0 0 f16x3; run 1 0 f16x3; run 1 10 f16x3
