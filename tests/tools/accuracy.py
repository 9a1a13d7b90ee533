#!/usr/bin/env python3
"""Measures the error of each precision path against the CPU oracle on a sample
of boards (test infrastructure; writes a JSON line)."""
import argparse, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib

ap = argparse.ArgumentParser()
ap.add_argument("--net", default="20x256"); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--sample", type=int, default=4); ap.add_argument("--bn", default="identity")
ap.add_argument("--positions", default="synthetic", choices=["synthetic", "game", "startpos"],
                help="synthetic: seeded random bitboards; game: distinct positions of random-playout games "
                     "(what self-play evaluates); startpos: the initial position in every slot (the benchmark's input)")
a = ap.parse_args()
nsg = importlib.import_module("nshogi-engine_amd")
blocks, ch = (int(x) for x in a.net.split("x"))
w = nsg.weights.make_random(blocks, ch, seed=0, bn=a.bn)
blob = nsg.weights.to_blob(w)
bb = {"synthetic": lambda: nsg.synth.random_batch(a.batch, 86, seed=1),
      "game": lambda: nsg.positions.game_positions(a.batch),
      "startpos": lambda: nsg.positions.startpos_batch(a.batch)}[a.positions]()
idx = np.linspace(0, a.batch - 1, a.sample).astype(int)
po, vo, do = oracle_lib.load().net(blob).evaluate_parallel(bb[idx])
res = {"net": a.net, "bn": a.bn, "positions": a.positions, "sample": int(len(idx)), "policy_abs_max_ref": float(np.abs(po).max()), "policy_std_ref": float(po.std())}
for prec in ("fp32", "f16x3", "f16m8", "f16m6", "fp16", "bf16"):
    ev = nsg.Evaluator(0, a.batch, 86, precision=prec); ev.load_memory(blob)
    p, v, d = ev.compute_blocking(bb)
    res[prec] = {"policy_max_abs_err": float(np.abs(p[idx] - po).max()),
                 "policy_rms_err": float(np.sqrt(((p[idx] - po) ** 2).mean())),
                 "value_max_abs_err": float(np.abs(v[idx] - vo).max()),
                 "draw_max_abs_err": float(np.abs(d[idx] - do).max())}
    ev.close()
print(json.dumps(res))
