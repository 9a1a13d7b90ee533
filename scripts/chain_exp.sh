#!/bin/bash
# chains experiment: evals/s for (chains, delay_us, batch); NSG_CHAIN_MIN_BATCH=64 enables chains at any batch
run() { NSG_CHAINS=$1 NSG_CHAIN_MIN_BATCH=64 NSG_CHAIN_DELAY_US=$2 python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --batch ${3:-512} --steps 20 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('chains=$1 delay=$2 batch=${3:-512}', round(d['value']), flush=True)"; }
for rep in 1 2; do
run 1 0; run 2 0; run 2 20; run 2 40; run 2 60
done
