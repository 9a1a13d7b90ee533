#!/bin/bash
# kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) vs the runtime's default, by batch
run() { HIP_FORCE_DEV_KERNARG=$1 python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --batch $2 --steps 60 --warmup 10 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('dev_kernarg=$1 batch=$2', round(d['value']), round(d['ms_per_step'],4), flush=True)"; }
for rep in 1 2; do for k in 0 1; do for b in 1 64 128 512; do run $k $b; done; done; done
