#!/usr/bin/env python3
"""A few device-resident forwards at one batch size (profiler workloads): scripts/one_forward.py [batch] [precision] [count]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
nsg = importlib.import_module("nshogi-engine_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "f16m6"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ev = nsg.Evaluator(0, B, 86, precision=prec)
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
ev.upload_features(nsg.positions.startpos_batch(B))
for _ in range(n):
    ev.forward_resident(B)
p, v, d = ev.download_outputs(B)
print("ok", float(p.max()), float(v[0]))
