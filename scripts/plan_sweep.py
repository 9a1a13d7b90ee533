#!/usr/bin/env python3
"""Device-resident evals/s, trunk time per forward (HIP events) and the launch plan by batch size for one net:
scripts/plan_sweep.py --net 10x192 --precision f16m6 --batches 1,8,16,17,32,64,128 [--positions startpos]
(one process; environment variables such as NSG_KSPLIT3=0 or NSG_TEAM_TRUNK=0 select the plans to compare)."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser(); ap.add_argument("--net", default="10x192"); ap.add_argument("--precision", default="f16m6")
ap.add_argument("--batches", default="1,2,4,8,16,17,24,32,48,64,85,96,128,256,512"); ap.add_argument("--seconds", type=float, default=0.5)
ap.add_argument("--positions", default="games", choices=["games", "startpos"])
a = ap.parse_args()
nsg = importlib.import_module("nshogi-engine_amd")
blocks, ch = (int(x) for x in a.net.split("x"))
batches = [int(x) for x in a.batches.split(",")]
bmax = max(batches)
ev = nsg.Evaluator(0, bmax, 86, precision=a.precision)
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(blocks, ch, seed=0, bn="identity")))
ev.upload_features(nsg.positions.game_positions(bmax, seed=9) if a.positions == "games" else nsg.positions.startpos_batch(bmax))
info = ev.info()
out = {}
for b in batches:
    for _ in range(3): ev.forward_resident(b)
    torch.cuda.synchronize()
    ev.profile_enable(True); ev.profile_read()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(8): ev.forward_resident(b)
        n += 8
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ev.profile_read(); ev.profile_enable(False)
    plan = ev.last_plan()
    rate = b * n / dt
    out[str(b)] = {"evals_per_sec": round(rate), "frac_mfma_peak": round(rate * info["flops_per_position"] / 2516.6e12, 4),
                   "trunk_us_per_forward": round(prof["trunk_ms_total"] / max(prof["forwards"], 1) * 1e3, 1),
                   "plan": "nb%d nf%d nw%d ms%d ks%d ss%d ch%d %s" % (plan["boards_per_group"], plan["fragments_per_wave"], plan["waves_per_group"],
                                                                   plan["row_split"], plan["k_split"], plan["slab_split"], plan["chains"], plan["trunk_precision"])}
    print(b, out[str(b)], flush=True)
print(json.dumps({"net": a.net, "precision": a.precision, "by_batch": out}))
