"""Oracle restatement of infer::Random / infer::Zero (src/infer/random.cc,
zero.cc) vs the known-answer vector G1 and vs this image's libstdc++, and the
product's CPU stand-in executors (host C++, no GPU needed) vs the oracle."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest


def test_g1_known_answers(oracle, golden_dir):
    g = json.load(open(f"{golden_dir}/random_g1.json"))
    st = oracle.mt(g["seed"])
    assert oracle.mt_next(st) == g["first_raw_mt19937_64"] == 2947667278772165694
    st = oracle.mt(g["seed"])
    p, w, d = oracle.random_compute(st, 2)
    flat = np.concatenate([p[0], w[:1], d[:1], p[1], w[1:], d[1:]]).view(np.uint32)
    for idx, bits in g["float_bits_by_draw_index"].items():
        assert f"{flat[int(idx)]:08x}" == bits, idx
    # the first four policy floats
    assert [f"{x:08x}" for x in p[0, :4].view(np.uint32)] == ["3e23a0df", "3f7dfd3a", "3d221321", "3f18f569"]
    # exactly one engine draw per float: 2 positions consume 2*2189 draws
    assert oracle.mt_next(st) == g["raw_after_4378_draws"]


def test_random_independent_of_batch_boundaries(oracle):
    a = oracle.random_compute(oracle.mt(0), 5)
    st = oracle.mt(0)
    parts = [oracle.random_compute(st, n) for n in (1, 3, 1)]
    for k in range(3):
        np.testing.assert_array_equal(a[k], np.concatenate([p[k] for p in parts]))


def test_random_range_and_zero(oracle):
    p, w, d = oracle.random_compute(oracle.mt(123), 3)
    for a in (p, w, d):
        assert (a >= 0).all() and (a < 1).all()
    p, w, d = oracle.zero_compute(4)
    assert not p.any() and not w.any() and not d.any()


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_oracle_vs_libstdcxx(oracle, tmp_path):
    """Cross-check against std::mt19937_64 + std::uniform_real_distribution<float>."""
    src = tmp_path / "x.cc"
    src.write_text(r'''
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <random>
int main(){ std::mt19937_64 r(20240203); std::uniform_real_distribution<float> d(0,1);
 for(int i=0;i<5000;++i){ float f=d(r); uint32_t u; std::memcpy(&u,&f,4); std::printf("%08x\n",u);} }
''')
    exe = tmp_path / "x"
    subprocess.check_call(["g++", "-O2", "-std=c++17", str(src), "-o", str(exe)])
    want = subprocess.check_output([str(exe)]).decode().split()
    st = oracle.mt(20240203)
    p, w, d = oracle.random_compute(st, 3)
    flat = np.concatenate([np.concatenate([p[i], w[i:i + 1], d[i:i + 1]]) for i in range(3)])
    got = [f"{x:08x}" for x in flat.view(np.uint32)[:5000]]
    assert got == want


def test_product_cpu_executors_match_oracle(nsg, oracle):
    """infer::Random / Zero / Nothing as shipped in libnsg.so (host code)."""
    if not os.path.exists(nsg.library_path()):
        pytest.skip("libnsg.so not built")
    ex = nsg.CpuExecutor("random", seed=0)
    p1, w1, d1 = ex.compute_blocking(3)
    p2, w2, d2 = ex.compute_blocking(2)  # the engine state carries across calls
    po, wo, do = oracle.random_compute(oracle.mt(0), 5)
    np.testing.assert_array_equal(np.concatenate([p1, p2]), po)
    np.testing.assert_array_equal(np.concatenate([w1, w2]), wo)
    np.testing.assert_array_equal(np.concatenate([d1, d2]), do)
    z = nsg.CpuExecutor("zero").compute_blocking(2)
    assert all(not a.any() for a in z)
    n = nsg.CpuExecutor("nothing").compute_blocking(2)  # leaves the buffers untouched
    assert all(np.isnan(a).all() for a in n)
