#!/usr/bin/env python3
"""Same-box comparison of two ROUNDS' libraries through the part of the C ABI both have
(raw ctypes: create / set_precision / load_memory / upload_features / forward_resident):
  scripts/ab_rounds.py <libA.so>:<precision> <libB.so>:<precision> [--batches 1,64,...]
e.g. the round-1 library in its default arithmetic (f16m8 = 4) against this round's (f16m6 = 5).
One process per library and round, alternating, median of three rounds per batch size."""
import argparse, ctypes, importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib_path, precision, batches):
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    nsg = importlib.import_module("nshogi-engine_amd")
    blob = nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity"))
    bmax = max(batches)
    bb = np.ascontiguousarray(nsg.synth.random_batch(bmax, 86, seed=9))
    lib = ctypes.CDLL(lib_path)
    vp, sz, i = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.nsg_create.argtypes = [i, i, i, ctypes.POINTER(vp)]
    lib.nsg_set_precision.argtypes = [vp, i]
    lib.nsg_load_memory.argtypes = [vp, ctypes.c_char_p, sz]
    lib.nsg_upload_features.argtypes = [vp, vp, sz]
    lib.nsg_forward_resident.argtypes = [vp, sz]
    lib.nsg_last_error.restype = ctypes.c_char_p
    h = vp()
    def ok(rc):
        assert rc == 0, lib.nsg_last_error()
    ok(lib.nsg_create(0, bmax, 86, ctypes.byref(h)))
    ok(lib.nsg_set_precision(h, precision))
    ok(lib.nsg_load_memory(h, blob, len(blob)))
    ok(lib.nsg_upload_features(h, bb.ctypes.data, bmax))
    out = {}
    for b in batches:
        for _ in range(3): ok(lib.nsg_forward_resident(h, b))
        torch.cuda.synchronize()
        n = max(8, min(400, int(0.25 * 150000 / max(b, 16))))
        t0 = time.perf_counter()
        for _ in range(n): ok(lib.nsg_forward_resident(h, b))
        torch.cuda.synchronize()
        out[b] = b * n / (time.perf_counter() - t0)
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), [int(x) for x in sys.argv[4].split(",")])
        sys.exit(0)
    ap = argparse.ArgumentParser(); ap.add_argument("a"); ap.add_argument("b"); ap.add_argument("--batches", default="1,16,64,96,128,192,256,512,1024")
    a = ap.parse_args()
    res = {a.a: [], a.b: []}
    for rnd in range(3):
        for spec in (a.a, a.b):
            path, prec = spec.rsplit(":", 1)
            r = subprocess.run([sys.executable, __file__, "--child", os.path.abspath(path), prec, a.batches], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            res[spec].append(json.loads(r.stdout.strip().splitlines()[-1]))
    print(f"{'batch':>6s} {a.a:>40s} {a.b:>40s}  ratio")
    for b in a.batches.split(","):
        m = [sorted(x[b] for x in res[s])[1] for s in (a.a, a.b)]
        print(f"{b:>6s} {m[0]:40.0f} {m[1]:40.0f}  {m[1] / m[0]:.3f}")
