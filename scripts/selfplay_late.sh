#!/bin/bash
# self-play leg at BASELINE config 4's shape over a full minute (games reach their middle and end games,
# where the df-pn call of judge gets expensive): host workers x solver threads
python - <<'PY'
import importlib, os, sys
sys.path.insert(0, os.getcwd())
nsg = importlib.import_module("nshogi-engine_amd")
open("/tmp/w.nsgw", "wb").write(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
PY
SECS=${SECS:-60}
for cfg in ${CFGS:-3:4 4:4 6:4 8:4}; do set -- ${cfg/:/ }
  nshogi-engine_amd/csrc/selfplay/selfplay --executor hip --weights /tmp/w.nsgw --gpu 0 --threads 1 --workers $1 --solver-threads $2 --games-per-group 128 --playouts 800 --seconds $SECS --seed 1 --precision 5 |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('workers $1 solvers $2', {k: round(d[k],2) for k in ('evals_per_sec','playouts_per_sec','moves_per_sec','games_per_sec','games_per_sec_window','avg_batch','cache_hit_ratio','avg_game_length','dfpn_nodes_per_move','await_ms_per_batch','host_ms_per_batch','batches_found_finished')}, flush=True)"
done
