#!/usr/bin/env python3
"""Device-resident evals/s by batch size for one or more precisions (one process, interleaved rounds:
scripts/batch_sweep.py f16m6 f16m8 [--net 20x256] [--batches 1,64,128,256,512])."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser(); ap.add_argument("precisions", nargs="+"); ap.add_argument("--net", default="20x256")
ap.add_argument("--batches", default="1,16,32,64,96,128,192,256,512,1024"); ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--lib", action="append", default=[], help="name=path of an alternative libnsg.so to compare (own process each)")
a = ap.parse_args()
nsg = importlib.import_module("nshogi-engine_amd")
blocks, ch = (int(x) for x in a.net.split("x"))
blob = nsg.weights.to_blob(nsg.weights.make_random(blocks, ch, seed=0, bn="identity"))
batches = [int(x) for x in a.batches.split(",")]
bmax = max(batches)
evs = {}
for p in a.precisions:
    ev = nsg.Evaluator(0, bmax, 86, precision=p); ev.load_memory(blob); ev.upload_features(nsg.positions.game_positions(bmax, seed=9)); evs[p] = ev
res = {p: {b: [] for b in batches} for p in a.precisions}
for r in range(a.rounds):
    for b in batches:
        for p in a.precisions:
            ev = evs[p]
            for _ in range(3): ev.forward_resident(b)
            torch.cuda.synchronize()
            n = max(8, min(400, int(0.25 * 150000 / max(b, 16))))
            t0 = time.perf_counter()
            for _ in range(n): ev.forward_resident(b)
            torch.cuda.synchronize()
            res[p][b].append(b * n / (time.perf_counter() - t0))
for b in batches:
    print(b, {p: round(sorted(res[p][b])[len(res[p][b]) // 2]) for p in a.precisions}, {p: evs[p].last_plan() for p in a.precisions[:1]} if False else "")
print(json.dumps({p: {str(b): sorted(v)[len(v) // 2] for b, v in res[p].items()} for p in a.precisions}))
