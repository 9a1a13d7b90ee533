#!/usr/bin/env python3
"""The plane-expansion kernel alone, a handful of launches at batch 512 -- the workload of the PMC passes behind
`roofline_extract.traffic` (scripts/pmc_extract.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
nsg = importlib.import_module("nshogi-engine_amd")
B, C = 512, 86
bb = nsg.positions.startpos_batch(B)
src = torch.from_numpy(bb.view(np.int64).copy()).cuda()
dst = torch.empty(B * C * 81, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for cf in (True, False):
    for _ in range(10):
        nsg.extract_bits(dst.data_ptr(), src.data_ptr(), B, C, cf, s)
torch.cuda.synchronize()
print("done")
