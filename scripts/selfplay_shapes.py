#!/usr/bin/env python3
"""Self-play leg at several engine shapes (threads per GPU x games per group): scripts/selfplay_shapes.py [seconds]"""
import importlib, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
nsg = importlib.import_module("nshogi-engine_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
blob = nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity"))
with tempfile.NamedTemporaryFile(suffix=".nsgw", delete=False) as f:
    f.write(blob); path = f.name
for threads, gpg, workers in ((1, 128, 8), (2, 64, 4), (2, 128, 4), (1, 256, 8), (4, 32, 2)):
    r = bench.selfplay_leg(path, 0, seconds, threads, "f16m6", games_per_group=gpg, workers=workers, solvers=4)
    keep = {k: r.get(k) for k in ("evals_per_sec", "games_per_sec_window", "avg_batch", "concurrent_games", "await_ms_per_batch", "host_ms_per_batch", "error")}
    print(f"threads {threads} games/group {gpg} workers {workers}: {json.dumps(keep)}", flush=True)
os.unlink(path)
