#!/usr/bin/env python3
"""Diagnostic (libnsg_diag.so built by `make diag_slabs`): cycles per slab of the kF16m8 main loop for workgroup 0 / wave 0,
averaged over the trunk layers -- shows where in the chunk the matrix pipe waits."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NSG_LIB"] = os.environ.get("NSG_DIAG_LIB") or os.path.join(ROOT, "nshogi-engine_amd", "csrc", "libnsg_diag.so")
nsg = importlib.import_module("nshogi-engine_amd")
blocks, ch, B = 20, 256, 512
ev = nsg.Evaluator(0, B, 86, precision=(sys.argv[1] if len(sys.argv) > 1 else "f16m8"))
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(blocks, ch, seed=0)))
lib = nsg.load_library()
lib.nsg_debug_stamps_enable.argtypes = [ctypes.c_void_p]; lib.nsg_debug_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ev.upload_features(nsg.synth.random_batch(B, 86, seed=1))
for _ in range(3): ev.forward_resident(B)
assert lib.nsg_debug_stamps_enable(ev._h) == 0
for _ in range(2): ev.forward_resident(B)
buf = np.zeros((2 * blocks, 4096 * 8), dtype=np.uint64)
assert lib.nsg_debug_stamps_read(ev._h, buf.ctypes.data) == 0
names = [n for t in range(9) for n in (f"A.m{t}", f"B.m{t}", f"X{t}")]
t = buf[2:, 2048:2048 + 4 * 32].astype(np.float64).reshape(-1, 4, 32)[:, :, :27]   # layer, chunk pair, slab
d = np.diff(t.reshape(t.shape[0], 4 * 27), axis=1)                                 # consecutive slab starts
d = np.concatenate([d, np.full((d.shape[0], 1), np.nan)], axis=1).reshape(-1, 4, 27)
print("ideal: m 704, X 1408 cycles")
print("slab  " + " ".join(f"{n:>6s}" for n in names))
for kp in range(4):
    print(f"pair={kp}  " + " ".join(f"{np.nanmean(d[:, kp, s]):6.0f}" for s in range(27)))
m = np.nanmean(d[:, 1:3, :], axis=(0, 1))
print("mean  " + " ".join(f"{x:6.0f}" for x in m), " pair total", round(float(m.sum())))
