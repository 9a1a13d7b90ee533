#!/usr/bin/env python3
"""Trunk / forward time by HIP events at the smallest batches (team trunk): scripts/team_profile.py [batches...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
nsg = importlib.import_module("nshogi-engine_amd")
ev = nsg.Evaluator(0, 8, 86, precision="f16m6")
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
ev.upload_features(nsg.positions.startpos_batch(8))
ev.profile_enable(True)
for b in [int(x) for x in sys.argv[1:]] or [1, 2, 8]:
    for _ in range(20): ev.forward_resident(b)
    torch.cuda.synchronize(); ev.profile_read()
    for _ in range(200): ev.forward_resident(b)
    torch.cuda.synchronize(); p = ev.profile_read()
    print(f"B={b}: trunk {p['trunk_ms_total'] / p['forwards'] * 1e3:.1f} us, forward {p['forward_ms_total'] / p['forwards'] * 1e3:.1f} us, plan {ev.last_plan()}")
