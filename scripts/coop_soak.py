#!/usr/bin/env python3
"""Soak of the cooperative trunk's hand-off (plain stores inside one XCD's L2, flags with XCC_ID): random batch sizes
and positions, every output compared bit for bit with the per-layer kernels' (NSG_COOP_TRUNK=0 evaluator), while a
second thread keeps another evaluator's per-layer forwards (batch 1024) on the device to make the load uneven.
scripts/coop_soak.py [seconds]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
nsg = importlib.import_module("nshogi-engine_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
blob = nsg.weights.to_blob(nsg.weights.make_random(6, 256, seed=3, bn="random"))
coop = nsg.Evaluator(0, 128, 86, precision="f16m6"); coop.load_memory(blob)
os.environ["NSG_COOP_TRUNK"] = "0"
per = nsg.Evaluator(0, 128, 86, precision="f16m6"); per.load_memory(blob)
del os.environ["NSG_COOP_TRUNK"]
noise = nsg.Evaluator(0, 1024, 86, precision="f16m6"); noise.load_memory(blob)
noise.upload_features(nsg.synth.random_batch(1024, 86, seed=5))
stop = False
def disturb():
    while not stop:
        for _ in range(4): noise.forward_resident(1024)
        noise.download_outputs(8)
        time.sleep(np.random.default_rng().uniform(0, 0.004))
t = threading.Thread(target=disturb); t.start()
rng = np.random.default_rng(11); pool = nsg.synth.random_batch(4096, 86, seed=7, garbage=True)
n = bad = 0; kinds = {}
t0 = time.time()
try:
    while time.time() - t0 < seconds:
        b = int(rng.integers(17, 129)); idx = rng.integers(0, 4096, b); bb = pool[idx]
        pc = coop.compute_blocking(bb); kinds[coop.last_launch_kind()[0]] = kinds.get(coop.last_launch_kind()[0], 0) + 1
        pp = per.compute_blocking(bb)
        ok = all(np.array_equal(x, y) for x, y in zip(pc, pp))
        n += 1; bad += 0 if ok else 1
        if not ok: print("MISMATCH at batch", b, float(np.abs(pc[0] - pp[0]).max()), flush=True)
finally:
    stop = True; t.join()
print({"forwards": n, "mismatches": bad, "launch_kinds": kinds, "team_stats": coop.team_stats(), "seconds": round(time.time() - t0, 1)})
sys.exit(1 if bad or coop.team_stats()["fallbacks"] else 0)
