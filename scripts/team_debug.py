#!/usr/bin/env python3
"""Team trunk against the per-layer kernels of the same arithmetic, by batch size: scripts/team_debug.py [blocks] [precision] [max batch]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
nsg = importlib.import_module("nshogi-engine_amd")
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
blob = nsg.weights.to_blob(nsg.weights.make_random(blocks, 256, seed=300 + blocks, bn="random"))
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
maxb = int(sys.argv[3]) if len(sys.argv) > 3 else 32
bb = nsg.synth.random_batch(maxb, 86, seed=301, garbage=True)
ev = nsg.Evaluator(0, maxb, 86, precision=prec); ev.load_memory(blob)
first = {n: ev.compute_blocking(bb[:n])[0] for n in (maxb, 1)}  # the team trunk is the first thing this process runs
os.environ["NSG_TEAM_TRUNK"] = "0"
old = nsg.Evaluator(0, maxb, 86, precision="f16x3"); old.load_memory(blob)
po, vo, do = old.compute_blocking(bb)
for n, p in first.items():
    err = np.abs(p - po[:n]).reshape(n, -1)
    print("first launches:", n, "max err per board", np.round(err.max(axis=1), 4).tolist())
    if err.max() > 1e-3:
        b = int(err.max(axis=1).argmax()); e = err[b].reshape(-1, 81)
        print("   board", b, "bad squares", np.nonzero(e.max(axis=0) > 1e-3)[0].tolist()[:81], "bad policy planes", np.nonzero(e.max(axis=1) > 1e-3)[0].tolist())
for n in [x for x in (8, 1, 2, 3, 12, 16, 17, 32, 25, 1, 8) if x <= maxb]:
    p, v, d = ev.compute_blocking(bb[:n])
    err = np.abs(p - po[:n]).reshape(n, -1)
    print(n, ev.last_plan()["row_split"], "max err per board", np.round(err.max(axis=1), 4).tolist())
    if err.max() > 1e-3:
        b = int(err.max(axis=1).argmax()); e = err[b].reshape(-1, 81)
        print("   board", b, "bad squares", np.nonzero(e.max(axis=0) > 1e-3)[0].tolist()[:40])
# the sequence of tests/test_gpu_evaluator.py::test_team_trunk_small_batches: one other board alone, then the full batch
# several times without a host wait in between
if maxb >= 8:
    for rep in range(3):
        ev.compute_blocking(bb[6:7])
        ev.upload_features(bb[:8])
        for i in range(5):
            ev.forward_resident(8)
        p = ev.download_outputs(8)[0]
        err = np.abs(p - po[:8]).reshape(8, -1)
        print("resident x5 without waits: max err per board", np.round(err.max(axis=1), 3).tolist())
