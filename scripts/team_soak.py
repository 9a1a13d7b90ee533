#!/usr/bin/env python3
"""Self-play on the team trunk (small leaf batches; an engine thread has two evaluators that share the device, SOAK_THREADS=2
adds a second engine thread): the same digest twice with one thread.
scripts/team_soak.py [seconds] [games per group]"""
import importlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
nsg = importlib.import_module("nshogi-engine_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0  # an upper bound: the run ends after --max-games
gpg = int(sys.argv[2]) if len(sys.argv) > 2 else 6
blob = nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity"))
with tempfile.NamedTemporaryFile(suffix=".nsgw", delete=False) as f:
    f.write(blob); path = f.name
out = []
for rep in range(2):
    r = subprocess.run([bench.SELFPLAY_BIN, "--executor", "hip", "--weights", path, "--gpu", "0", "--threads", os.environ.get("SOAK_THREADS", "1"), "--workers", os.environ.get("SOAK_WORKERS", "1"),
                        "--solver-threads", os.environ.get("SOAK_SOLVERS", "0"), "--games-per-group", str(gpg), "--playouts", "200", "--max-games", "24",
                        "--seconds", str(seconds), "--seed", "7", "--precision", "5"], capture_output=True, text=True, timeout=seconds * 3 + 300)
    if r.returncode != 0:
        print("selfplay failed:", (r.stderr or r.stdout)[-500:]); sys.exit(1)
    j = json.loads(r.stdout.strip().split("\n")[-1])
    out.append(j)
    print({k: j.get(k) for k in ("evals_per_sec", "avg_batch", "games_finished", "moves", "digest", "seconds")}, flush=True)
os.unlink(path)
print("same digest:", out[0].get("digest") == out[1].get("digest"), "same games:", out[0].get("games_finished") == out[1].get("games_finished"))
