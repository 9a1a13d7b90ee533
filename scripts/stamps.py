#!/usr/bin/env python3
"""Diagnostic: where a trunk-conv workgroup spends its cycles (libnsg_diag.so).
Never used for timing claims: the stamped build is slower than the product build."""
import argparse, ctypes, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NSG_LIB"] = os.environ.get("NSG_DIAG_LIB") or os.path.join(ROOT, "nshogi-engine_amd", "csrc", "libnsg_diag.so")
ap = argparse.ArgumentParser(); ap.add_argument("--precision", default="f16x3"); ap.add_argument("--net", default="20x256")
ap.add_argument("--batch", type=int, default=512); a = ap.parse_args()
nsg = importlib.import_module("nshogi-engine_amd")
blocks, ch = (int(x) for x in a.net.split("x"))
ev = nsg.Evaluator(0, a.batch, 86, precision=a.precision)
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(blocks, ch, seed=0)))
lib = nsg.load_library()
lib.nsg_debug_stamps_enable.argtypes = [ctypes.c_void_p]; lib.nsg_debug_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
bb = nsg.synth.random_batch(a.batch, 86, seed=1); ev.upload_features(bb)
for _ in range(3): ev.forward_resident(a.batch)
assert lib.nsg_debug_stamps_enable(ev._h) == 0
for _ in range(2): ev.forward_resident(a.batch)
buf = np.zeros((2 * blocks, 4096, 8), dtype=np.uint64)
assert lib.nsg_debug_stamps_read(ev._h, buf.ctypes.data) == 0
res = []
for l in range(2 * blocks):
    s = buf[l]; ok = s[:, 0] != 0; s = s[ok].astype(np.float64)
    t0 = s[:, 0].min()
    d = {"layer": l, "wgs": int(ok.sum()),
         "start_spread": float(s[:, 0].max() - t0),
         "prologue": float(np.median(s[:, 1] - s[:, 0])),
         "prologue_setup_and_issue": float(np.median(s[:, 4] - s[:, 0])) if s[:, 4].any() else None,  # K-split kernels only
         "prologue_clear_lds": float(np.median(s[:, 5] - s[:, 4])) if s[:, 4].any() else None,
         "prologue_wait_tiles_and_stage": float(np.median(s[:, 1] - s[:, 5])) if s[:, 4].any() else None,
         "mainloop": float(np.median(s[:, 2] - s[:, 1])),
         "epilogue": float(np.median(s[:, 3] - s[:, 2])),
         "epilogue_max": float((s[:, 3] - s[:, 2]).max()),
         "total_med": float(np.median(s[:, 3] - s[:, 0])),
         "first_start_to_last_end": float(s[:, 3].max() - t0),
         "mhz": float(np.median((s[:, 3] - s[:, 0]) / np.maximum(s[:, 6] - s[:, 7], 1)) * 100.0)}
    res.append(d)
print(json.dumps(res[2:6], indent=0))
agg = {k: float(np.mean([r[k] for r in res[2:]])) for k in res[0] if k not in ("layer", "wgs") and res[2][k] is not None}
print("mean over layers:", json.dumps(agg))
