#!/bin/bash
# A/B of two builds of libnsg on one box: scripts/ab.sh <libA> <libB> [bench args...]
# Alternates the two libraries so clock / box drift cancels; prints evals/s and the
# trunk conv's average launch time for each precision.
A=$1; B=$2; shift 2
for rep in 1 2; do
  for lib in "$A" "$B"; do
    for prec in ${PRECS:-f16x3 bf16 fp32}; do
      NSG_LIB=$lib python bench.py --precision $prec --selfplay-seconds 0 --no-cpu-baseline --no-host-path "$@" 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', '$prec', round(d['value']), round(d['roofline']['avg_launch_ms'],4), flush=True)"
    done
  done
done
