#!/bin/bash
# chains experiment: evals/s for (chains, min batch, batch)
run() { NSG_CHAINS=$1 NSG_CHAIN_MIN_BATCH=64 NSG_CHAIN_DELAY_US=$2 python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --batch ${3:-512} --steps 10 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('chains=$1 delay=$2 batch=${3:-512}', round(d['value']), flush=True)"; }
for b in 640 768 1024 1536 2048 4096; do
run 1 0 $b; run 2 0 $b; run 3 0 $b; run 4 0 $b
done
