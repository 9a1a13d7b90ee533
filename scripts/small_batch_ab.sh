#!/bin/bash
# small / mid batches, two builds alternated on one box: scripts/small_batch_ab.sh <libA> <libB>
for rep in 1 2; do for lib in "$1" "$2"; do for b in ${BATCHES:-33 48 64 96 128}; do
  NSG_LIB=$lib python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --batch $b --net ${NET:-20x256} --steps 60 --warmup 10 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib batch=$b', round(d['value']), round(d['ms_per_step'],4), flush=True)"
done; done; done
