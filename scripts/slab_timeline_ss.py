import ctypes, importlib, os, sys
import numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT)
os.environ["NSG_LIB"]=os.environ["NSG_DIAG_LIB"]
nsg = importlib.import_module("nshogi-engine_amd")
B=int(sys.argv[1])
ev = nsg.Evaluator(0, B, 86, precision="f16m6")
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0)))
lib = nsg.load_library()
lib.nsg_debug_stamps_enable.argtypes=[ctypes.c_void_p]; lib.nsg_debug_stamps_read.argtypes=[ctypes.c_void_p, ctypes.c_void_p]
ev.upload_features(nsg.synth.random_batch(B, 86, seed=1))
for _ in range(3): ev.forward_resident(B)
assert lib.nsg_debug_stamps_enable(ev._h)==0
for _ in range(2): ev.forward_resident(B)
buf=np.zeros((40, 4096*8), dtype=np.uint64)
assert lib.nsg_debug_stamps_read(ev._h, buf.ctypes.data)==0
t=buf[2:, 2048:2048+4*32].astype(np.float64).reshape(-1,4,32)
# part 0 of SS=4 runs real slabs: M0 X0 M1 X1 M2 X2 X3 = 0,2,1,5,3,8,11
order=[0,2,1,5,3,8,11] if len(sys.argv)<3 else [int(x) for x in sys.argv[2].split(",")]
L=t.mean(axis=0)  # pair, slab
for kp in range(4):
    ts=[L[kp][s] for s in order]
    print("pair",kp,"slab starts rel:",[int(x-ts[0]) for x in ts], "next pair start:", int(L[kp+1][order[0]]-ts[0]) if kp<3 else None)
