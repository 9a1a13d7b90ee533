#!/usr/bin/env python3
"""Time nsg_load_memory (BN folding + fragment packing + upload) per precision."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nsg = importlib.import_module("nshogi-engine_amd")
for net in ((20, 256), (40, 384)):
    blob = nsg.weights.to_blob(nsg.weights.make_random(*net, seed=0))
    for prec in ("fp32", "f16x3", "f16m8"):
        ev = nsg.Evaluator(0, 512, 86, precision=prec)
        t0 = time.perf_counter(); ev.load_memory(blob); dt = time.perf_counter() - t0
        print(f"{net[0]}x{net[1]} {prec}: load {dt:.2f} s", flush=True)
        ev.close()
