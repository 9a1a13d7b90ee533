#!/bin/bash
# 256 concurrent games per GPU (BASELINE config 4) as engines x 2 groups x games, with W host threads
# advancing each engine's games between two batches: which shape feeds the evaluator best
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
nsg = importlib.import_module("nshogi-engine_amd")
open("/tmp/w.nsgw", "wb").write(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
PY
SECS=${SECS:-20}
for cfg in "1 1 128" "1 2 128" "1 3 128" "1 4 128" "1 6 128" "2 2 64" "2 1 64" "1 4 256"; do set -- $cfg
  nshogi-engine_amd/csrc/selfplay/selfplay --executor hip --weights /tmp/w.nsgw --gpu 0 --threads $1 --workers $2 --games-per-group $3 --playouts 800 --seconds $SECS --seed 1 --precision 4 |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('engines $1 workers $2 group $3', {k: round(d[k],2) for k in ('evals_per_sec','playouts_per_sec','moves_per_sec','games_per_sec_window','avg_batch','cache_hit_ratio')}, flush=True)"
done
