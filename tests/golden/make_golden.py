"""Generates the committed golden fixtures (run from the repo root:
`python tests/golden/make_golden.py`).

The reference holds no golden vectors for this path and
cannot be built here, so these fixtures pin the build's own restatements:
  extract_g2.npz   -- branch-coverage feature bitboards and
                      their planes from the independent numpy restatement
                      `synth.expand_reference` (NOT from the C oracle; the
                      C oracle and the HIP kernels are both checked against it)
  random_g1.json   -- known-answer vector of infer::Random(0),
                      measured with this image's libstdc++ std::mt19937_64 +
                      std::uniform_real_distribution<float>
  net_tiny.npz     -- a 2-block x 64-channel synthetic net (seed 7, random BN),
                      6 synthetic positions and the C oracle's outputs for them
"""
import importlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))

nsg = importlib.import_module("nshogi-engine_amd")
synth = nsg.synth


def g2_bitboards():
    C = 86
    cases = []

    def plane(squares, rotate, value):
        bits = np.zeros(81, dtype=bool)
        bits[list(squares)] = True
        return synth.pack(bits, rotate, np.float32(value))

    single = [plane([s], r, v) for s in (0, 62, 63, 80) for r in (False, True)
              for v in (1.0, 0.5, 0.0)]
    full = [plane(range(81), r, 1.0) for r in (False, True)]
    lo_only = [plane(range(63), r, 1.0) for r in (False, True)]
    hi_only = [plane(range(63, 81), r, 0.25) for r in (False, True)]
    neg = [plane([5, 40, 79], False, -2.5), plane([5, 40, 79], True, 3.0e-39)]  # sign bit, subnormal
    hand = single + full + lo_only + hi_only + neg
    pos = np.zeros((2, C, 2), dtype=np.uint64)
    for i, p in enumerate(hand[:2 * C]):
        pos[i // C, i % C] = p
    cases.append(pos)
    cases.append(synth.random_batch(5, C, seed=11, garbage=True))
    # all-ones words: every ignored bit set, value = all-ones pattern (a NaN)
    ones = np.full((1, C, 2), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    cases.append(ones)
    return np.concatenate(cases, axis=0)


def main():
    bb = g2_bitboards()
    np.savez_compressed(os.path.join(HERE, "extract_g2.npz"), bitboards=bb,
                        nchw_bits=synth.expand_reference(bb, True).view(np.uint32),
                        nhwc_bits=synth.expand_reference(bb, False).view(np.uint32))

    # G1 straight from libstdc++
    src = r'''
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
int main() {
    std::mt19937_64 probe(0);
    std::printf("%llu\n", (unsigned long long)probe());
    std::mt19937_64 rng(0);
    std::uniform_real_distribution<float> dist(0, 1);
    for (int i = 0; i < 2189 * 2; ++i) {
        float f = dist(rng); uint32_t u; std::memcpy(&u, &f, 4);
        if (i < 8 || (i >= 2187 && i < 2193)) std::printf("%d %08x\n", i, u);
    }
    std::printf("next %llu\n", (unsigned long long)rng());
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as td:
        cc = os.path.join(td, "g1.cc")
        open(cc, "w").write(src)
        exe = os.path.join(td, "g1")
        subprocess.check_call(["g++", "-O2", "-std=c++17", cc, "-o", exe])
        lines = subprocess.check_output([exe]).decode().split("\n")
    first_raw = int(lines[0])
    floats = {}
    nxt = None
    for ln in lines[1:]:
        if ln.startswith("next"):
            nxt = int(ln.split()[1])
        elif ln.strip():
            i, h = ln.split()
            floats[int(i)] = h
    json.dump({"seed": 0, "first_raw_mt19937_64": first_raw, "float_bits_by_draw_index": floats,
               "raw_after_4378_draws": nxt,
               "note": "draw index i = position*2189 + j; j<2187 policy, 2187 win, 2188 draw"},
              open(os.path.join(HERE, "random_g1.json"), "w"), indent=1)

    import oracle_lib
    o = oracle_lib.load()
    w = nsg.weights.make_random(2, 64, seed=7, bn="random")
    blob = nsg.weights.to_blob(w)
    pos = synth.random_batch(6, 86, seed=3)
    p, v, d = o.net(blob).evaluate(pos)
    np.savez_compressed(os.path.join(HERE, "net_tiny.npz"), weights_seed=7, blocks=2, channels=64,
                        bitboards=pos, policy=p, value=v, draw=d)
    print("golden fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
