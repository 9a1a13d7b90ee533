#!/bin/bash
# Same-box A/B of two builds of libnsg.so: scripts/ab_libs.sh <old.so> <new.so> [batch_sweep args...]
# alternates the two libraries (own process each), two rounds, prints evals/s by batch
OLD=$1; NEW=$2; shift 2
for round in 1 2; do
  for which in old new; do
    lib=$OLD; [ $which = new ] && lib=$NEW
    echo "== round $round $which"
    NSG_LIB=$PWD/$lib python scripts/batch_sweep.py "$@" 2>/dev/null | tail -1
  done
done
