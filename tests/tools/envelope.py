#!/usr/bin/env python3
"""Accuracy envelope of the MX trunk formats (test infrastructure; prints JSON lines).
20x256 net at batch 512, ALL boards, against the f32-equivalent F16X3 evaluator (5e-6 from the CPU
oracle): the stem's BN gain is multiplied by g (activations and logits scale with it) and the
convolution weights of one block are spread over `spread` binades.  F16M8 is run with its load-time
guard off (NSG_M8_GUARD=0) to show where its fixed-scale window ends, and with the guard on."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
nsg = importlib.import_module("nshogi-engine_amd")
B = 512
bb = nsg.positions.game_positions(B, seed=3)
for bn in ("identity", "random"):
    for spread in (0, 8):
        for g in (1, 2, 4, 8, 16, 64, 256):
            w = nsg.weights.make_random(20, 256, seed=0, bn=bn)
            w["stem_bn"][0] *= float(g)
            if spread:
                rng = np.random.default_rng(1)
                for k in ("b3_w1", "b11_w2"):
                    w[k] = (w[k] * np.exp2(rng.integers(-spread, 1, size=w[k].shape[:1])).reshape(-1, 1, 1, 1)).astype(np.float32)
            blob = nsg.weights.to_blob(w)
            ref = nsg.Evaluator(0, B, 86, precision="f16x3"); ref.load_memory(blob)
            pr, vr, dr = ref.compute_blocking(bb); ref.close()
            row = {"bn": bn, "weight_binade_spread": spread, "stem_gain": g, "logit_max": float(np.abs(pr).max())}
            for name, prec, guard in (("f16m6", "f16m6", "1"), ("f16m8_guard_off", "f16m8", "0"), ("f16m8", "f16m8", "1")):
                os.environ["NSG_M8_GUARD"] = guard
                ev = nsg.Evaluator(0, B, 86, precision=prec); ev.load_memory(blob)
                p, v, d = ev.compute_blocking(bb)
                info = ev.info()
                row[name] = {"policy_err_over_logit_max": float(np.abs(p - pr).max() / max(1.0, np.abs(pr).max())),
                             "value_err": float(np.abs(v - vr).max()), "ran_as": ev.last_plan()["trunk_precision"]}
                row["activation_bound_estimate"] = info["activation_bound_estimate"]
                ev.close()
            print(json.dumps(row), flush=True)
