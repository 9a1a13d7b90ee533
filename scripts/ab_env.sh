#!/bin/bash
# Same-box A/B of one environment variable: scripts/ab_env.sh VAR A B [batch_sweep args...]
VAR=$1; A=$2; B=$3; shift 3
for round in 1 2 3; do
  for v in "$A" "$B"; do
    echo "== round $round $VAR=$v"
    env $VAR=$v python scripts/batch_sweep.py "$@" 2>/dev/null | tail -1
  done
done
