#!/usr/bin/env python3
"""Diagnostic (libnsg_diag.so): where the workgroups of a PERSISTENT trunk launch (trunk kernel, cooperative trunk) spend
their cycles, per layer: wait between two layers (end of the epilogue -> entry of the next layer: the flag hand-off of the
cooperative trunk), prologue parts, main loop (+ K reduction), epilogue.  Never used for timing claims.
scripts/stamps_coop.py --batch 64 [--net 20x256]"""
import argparse, ctypes, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NSG_LIB"] = os.environ.get("NSG_DIAG_LIB") or os.path.join(ROOT, "nshogi-engine_amd", "csrc", "libnsg_diag.so")
ap = argparse.ArgumentParser(); ap.add_argument("--precision", default="f16m6"); ap.add_argument("--net", default="20x256")
ap.add_argument("--batch", type=int, default=64); a = ap.parse_args()
nsg = importlib.import_module("nshogi-engine_amd")
blocks, ch = (int(x) for x in a.net.split("x"))
ev = nsg.Evaluator(0, a.batch, 86, precision=a.precision)
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(blocks, ch, seed=0)))
lib = nsg.load_library()
lib.nsg_debug_stamps_enable.argtypes = [ctypes.c_void_p]; lib.nsg_debug_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ev.upload_features(nsg.positions.game_positions(a.batch, seed=9))
for _ in range(3): ev.forward_resident(a.batch)
assert lib.nsg_debug_stamps_enable(ev._h) == 0
for _ in range(2): ev.forward_resident(a.batch)
buf = np.zeros((2 * blocks, 4096, 8), dtype=np.uint64)
assert lib.nsg_debug_stamps_read(ev._h, buf.ctypes.data) == 0
print("plan", ev.last_plan(), "launch", ev.last_launch_kind() if hasattr(ev, "last_launch_kind") else "")
rows = []
for l in range(2, 2 * blocks - 1):
    s, n = buf[l].astype(np.float64), buf[l + 1].astype(np.float64)
    ok = (s[:, 0] != 0) & (n[:, 0] != 0)
    s, n = s[ok], n[ok]
    mhz = np.median((s[:, 3] - s[:, 0]) / np.maximum(s[:, 6] - s[:, 7], 1)) * 100.0
    rows.append({"wgs": int(ok.sum()), "mhz_memtime": float(mhz),
                 "issue_tiles": float(np.median(s[:, 4] - s[:, 0])), "clear_lds": float(np.median(s[:, 5] - s[:, 4])),
                 "wait_tiles_and_stage": float(np.median(s[:, 1] - s[:, 5])), "loop_and_k_sum": float(np.median(s[:, 2] - s[:, 1])),
                 "epilogue": float(np.median(s[:, 3] - s[:, 2])), "layer": float(np.median(s[:, 3] - s[:, 0])),
                 "between_layers_med": float(np.median(n[:, 0] - s[:, 3])), "between_layers_max": float((n[:, 0] - s[:, 3]).max()),
                 "period": float(np.median(n[:, 0] - s[:, 0]))})
agg = {k: round(float(np.mean([r[k] for r in rows])), 1) for k in rows[0]}
print("mean over layers (s_memtime ticks, 100 MHz: 1 tick = 10 ns):" if agg["mhz_memtime"] < 150 else "mean over layers (cycles):", json.dumps(agg))
