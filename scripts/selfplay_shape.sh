#!/bin/bash
# 256 concurrent games per GPU (BASELINE config 4) as threads x 2 groups x games: which shape feeds the evaluator best
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
nsg = importlib.import_module("nshogi-engine_amd")
open("/tmp/w.nsgw", "wb").write(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
PY
for cfg in "2 64" "1 128" "4 32" "2 64" "1 128"; do set -- $cfg
  nshogi-engine_amd/csrc/selfplay/selfplay --executor hip --weights /tmp/w.nsgw --gpu 0 --threads $1 --games-per-group $2 --playouts 800 --seconds 15 --seed 1 --precision 4 |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('threads $1 group $2', {k: round(d[k],2) for k in ('evals_per_sec','playouts_per_sec','moves_per_sec','avg_batch','cache_hit_ratio')}, flush=True)"
done
