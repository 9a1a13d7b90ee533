#!/bin/bash
# self-play leg with the df-pn solver in judge on (reference default, 100000 nodes) and off
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
nsg = importlib.import_module("nshogi-engine_amd")
open("/tmp/w.nsgw", "wb").write(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
PY
for cfg in "2 64" "4 32" "8 64"; do set -- $cfg
  for dn in 100000 10000 0; do
  nshogi-engine_amd/csrc/selfplay/selfplay --executor hip --weights /tmp/w.nsgw --gpu 0 --threads $1 --games-per-group $2 --playouts 800 --seconds 15 --seed 1 --precision 4 --dfpn-nodes $dn --teacher /tmp/t.nsgt |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('threads $1 group $2 dfpn $dn', {k: round(d[k],2) for k in ('evals_per_sec','playouts_per_sec','moves_per_sec','games_per_sec','avg_batch','cache_hit_ratio','dfpn_mates','dfpn_nodes_per_move','teacher_records','avg_game_length')}, flush=True)"
  done
done
