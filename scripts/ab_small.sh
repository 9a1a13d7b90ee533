#!/bin/bash
# A/B of two libnsg builds at small batches: scripts/ab_small.sh <libA> <libB>
A=$1; B=$2
run() { NSG_LIB=$1 python bench.py --selfplay-seconds 0 --no-cpu-baseline --no-host-path --net $2 --batch $3 --steps 40 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1 $2 B=$3', round(d['value']), round(d['ms_per_step'],3), flush=True)"; }
for lib in "$A" "$B" "$A" "$B"; do
  run $lib 20x256 1; run $lib 20x256 32; run $lib 20x256 64; run $lib 10x192 64; run $lib 20x256 128
done
