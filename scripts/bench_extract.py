#!/usr/bin/env python3
"""HBM roofline of the plane-expansion kernel (nsg_extract_bits, the K1/K2 replacement).
Algorithmic bytes per position: 1376 read + 27 864 written."""
import argparse, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

def measure(nsg, batch, channels_first, iters=50):
    C = 86
    bb = nsg.synth.random_batch(min(batch, 4096), C, seed=1)
    reps = (batch + bb.shape[0] - 1) // bb.shape[0]
    bb = np.concatenate([bb] * reps)[:batch]
    src = torch.from_numpy(bb.view(np.int64).copy()).cuda()
    dst = torch.empty(batch * C * 81, dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        nsg.extract_bits(dst.data_ptr(), src.data_ptr(), batch, C, channels_first, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        nsg.extract_bits(dst.data_ptr(), src.data_ptr(), batch, C, channels_first, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = batch * (1376 + 27864)
    return {"batch": batch, "layout": "NCHW" if channels_first else "NHWC", "ms": ms,
            "GB_per_s": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000.0,
            "positions_per_s": batch / ms * 1e3}

if __name__ == "__main__":
    nsg = importlib.import_module("nshogi-engine_amd")
    out = [measure(nsg, b, cf) for b in (512, 1024, 16384, 65535) for cf in (True, False)]
    print(json.dumps(out))
