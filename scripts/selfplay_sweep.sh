#!/bin/bash
# self-play leg at 256 concurrent games per GPU: threads x games-per-group sweep
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
nsg = importlib.import_module("nshogi-engine_amd")
open("/tmp/w.nsgw", "wb").write(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
PY
for cfg in "2 64" "4 32" "8 16" "4 64" "8 32" "8 64"; do set -- $cfg
  for ms in 1 0; do
  nshogi-engine_amd/csrc/selfplay/selfplay --executor hip --weights /tmp/w.nsgw --gpu 0 --threads $1 --games-per-group $2 --playouts 800 --seconds 12 --seed 1 --precision 4 --mate-search $ms |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('threads $1 group $2 mate $ms', {k: round(d[k],1) for k in ('evals_per_sec','playouts_per_sec','moves_per_sec','avg_batch','cache_hit_ratio','mates_found')}, flush=True)"
  done
done
