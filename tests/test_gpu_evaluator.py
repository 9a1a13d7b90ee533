"""The HIP evaluator behind the C ABI vs the CPU oracle (same seeded inputs),
vs the committed golden fixture, and size-independent properties at
BASELINE.json's full sizes.  Tolerance for the network outputs is the
north_star's 1e-3 (fp32 path); reduced-precision paths state their own."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3  # north_star: "within 1e-3 fp32"


def make(nsg, blocks, channels, batch_max, precision="fp32", seed=0, bn="random"):
    w = nsg.weights.make_random(blocks, channels, seed=seed, bn=bn)
    blob = nsg.weights.to_blob(w)
    ev = nsg.Evaluator(0, batch_max, 86, precision=precision)
    ev.load_memory(blob)
    return ev, blob


def check(a, b, tol):
    for x, y, name in zip(a, b, ("policy", "value", "draw")):
        err = float(np.abs(x - y).max())
        assert err <= tol, f"{name}: max abs err {err} > {tol}"


def test_golden_fixture(nsg, golden_dir):
    g = np.load(f"{golden_dir}/net_tiny.npz")
    ev, _ = make(nsg, int(g["blocks"]), int(g["channels"]), 8, seed=int(g["weights_seed"]))
    out = ev.compute_blocking(g["bitboards"])
    check(out, (g["policy"], g["value"], g["draw"]), TOL)
    assert np.isfinite(out[0]).all()


@pytest.mark.parametrize("blocks,channels", [(1, 64), (2, 128), (2, 192), (1, 256), (1, 384)])
def test_fp32_vs_oracle_shapes(nsg, oracle, blocks, channels):
    ev, blob = make(nsg, blocks, channels, 16, seed=channels + blocks)
    net = oracle.net(blob)
    bb = nsg.synth.random_batch(9, 86, seed=5, garbage=True)
    check(ev.compute_blocking(bb), net.evaluate(bb), 2e-4)
    trunk = ev.download_trunk(9)
    _, _, _, t_ref = net.forward_planes(oracle.extract_bits(bb), want_trunk=True)
    assert float(np.abs(trunk - t_ref).max()) < 2e-4


@pytest.mark.parametrize("batch", [1, 2, 3, 31, 64, 65])
def test_fp32_batch_sizes(nsg, oracle, batch):
    """Variable BatchSize per call down to 1, odd sizes, the maximum."""
    ev, blob = make(nsg, 2, 64, 65, seed=3)
    bb = nsg.synth.random_batch(batch, 86, seed=batch)
    check(ev.compute_blocking(bb), oracle.net(blob).evaluate(bb), 2e-4)


def test_config2_net_10x192_batch64(nsg, oracle):
    """BASELINE config 2's net and batch; the oracle checks a sample of boards."""
    ev, blob = make(nsg, 10, 192, 64, seed=1, bn="identity")
    bb = nsg.synth.random_batch(64, 86, seed=2)
    p, v, d = ev.compute_blocking(bb)
    idx = [0, 1, 31, 63]
    po, vo, do = oracle.net(blob).evaluate(bb[idx])
    check((p[idx], v[idx], d[idx]), (po, vo, do), TOL)


@pytest.mark.parametrize("channels,batch", [(192, 256), (256, 130), (256, 300), (128, 512), (384, 96)])
def test_tile_plans_across_batch_sizes(nsg, oracle, channels, batch):
    """Every tile plan the heuristic can pick (1/2 boards per workgroup, 1/2/4
    fragments per wave) gives the same answers: sample boards vs the oracle."""
    ev, blob = make(nsg, 1, channels, batch, seed=channels + batch)
    bb = nsg.synth.random_batch(batch, 86, seed=batch)
    p, v, d = ev.compute_blocking(bb)
    idx = [0, 1, batch // 2, batch - 2, batch - 1]
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), 2e-4)


def test_outputs_in_range_and_finite(nsg):
    ev, _ = make(nsg, 2, 64, 32, seed=9)
    p, v, d = ev.compute_blocking(nsg.synth.random_batch(32, 86, seed=1))
    assert np.isfinite(p).all() and np.isfinite(v).all() and np.isfinite(d).all()
    assert ((v >= 0) & (v <= 1)).all() and ((d >= 0) & (d <= 1)).all()


def test_batch_composition_independence_full_size(nsg):
    """Config 3 (20x256, B=512): every position's result is independent of what
    else is in the batch and of its slot -- bit-exact on the fp32 path."""
    ev, _ = make(nsg, 20, 256, 512, seed=4, bn="identity")
    bb = nsg.synth.random_batch(512, 86, seed=8)
    p, v, d = ev.compute_blocking(bb)
    assert np.isfinite(p).all()
    perm = np.random.default_rng(0).permutation(512)
    p2, v2, d2 = ev.compute_blocking(bb[perm])
    np.testing.assert_array_equal(p2, p[perm])
    np.testing.assert_array_equal(v2, v[perm])
    p3, v3, d3 = ev.compute_blocking(bb[:37])   # a short, odd batch
    np.testing.assert_allclose(p3, p[:37], atol=1e-5, rtol=0)  # tile shape may differ
    np.testing.assert_allclose(v3, v[:37], atol=1e-6, rtol=0)
    # replicated positions give replicated outputs (the reference benchmark's input)
    rep = nsg.synth.random_batch(512, 86, seed=8, distinct=False)
    pr, vr, dr = ev.compute_blocking(rep)
    assert (pr == pr[0]).all() and (vr == vr[0]).all() and (dr == dr[0]).all()


def test_f16x3_full_size_properties(nsg, oracle):
    """The default precision at BASELINE's full size (20x256, B=512): a sample of
    boards against the oracle at the north_star tolerance, and bit-exact
    independence from batch composition and slot."""
    ev, blob = make(nsg, 20, 256, 512, precision="f16x3", seed=4, bn="identity")
    bb = nsg.synth.random_batch(512, 86, seed=8)
    p, v, d = ev.compute_blocking(bb)
    idx = [0, 255, 511]
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), 1e-4)  # 10x tighter than 1e-3
    perm = np.random.default_rng(1).permutation(512)
    p2, v2, d2 = ev.compute_blocking(bb[perm])
    np.testing.assert_array_equal(p2, p[perm])
    np.testing.assert_array_equal(v2, v[perm])
    np.testing.assert_array_equal(d2, d[perm])
    assert ((v >= 0) & (v <= 1)).all() and ((d >= 0) & (d <= 1)).all() and np.isfinite(p).all()


@pytest.mark.parametrize("mx", ["f16m8", "f16m6"])
@pytest.mark.parametrize("blocks,channels,batch", [(1, 64, 3), (2, 128, 9), (2, 192, 5), (3, 256, 70), (1, 384, 4)])
def test_f16m8_vs_oracle(nsg, oracle, monkeypatch, blocks, channels, batch, mx):
    """f16 main term + MX correction terms (trunk convolutions; f16m8: e4m3 operands with fixed
    scales, f16m6: e2m3 operands with one E8M0 scale per 32 channels): outputs and the trunk
    activation against the oracle, held to the north_star's 1e-3 (measured ~1e-4)."""
    monkeypatch.setenv("NSG_CONV_NFRAG", "4")  # full tiles even at these small batches
    ev, blob = make(nsg, blocks, channels, batch, precision=mx, seed=40 + channels)
    net = oracle.net(blob)
    bb = nsg.synth.random_batch(batch, 86, seed=11, garbage=True)
    out = ev.compute_blocking(bb)
    n = min(batch, 6)
    ref = net.evaluate(bb[:n])
    check((out[0][:n], out[1][:n], out[2][:n]), ref, TOL)
    assert ev.last_plan()["trunk_precision"] == mx
    trunk = ev.download_trunk(n)
    _, _, _, t_ref = net.forward_planes(oracle.extract_bits(bb[:n]), want_trunk=True)
    assert float(np.abs(trunk - t_ref).max()) < TOL * max(1.0, float(np.abs(t_ref).max()))


def test_f16m8_small_batches_run_as_f16x3(nsg, oracle):
    """An f16m8 evaluator keeps the trunk in both forms: where no f16m8 plan fits a small batch
    (here 192 channels: three chunk pairs; a three-way K split measured 3-5 % slower than these) it
    takes the f16x3 small-tile kernels (f32-equivalent), larger batches the MX path."""
    ev, blob = make(nsg, 2, 192, 300, precision="f16m8", seed=44)
    net = oracle.net(blob)
    bb = nsg.synth.random_batch(300, 86, seed=12)
    p, v, d = ev.compute_blocking(bb[:16])
    assert ev.last_plan()["trunk_precision"] == "f16x3"
    check((p[:4], v[:4], d[:4]), net.evaluate(bb[:4]), 1e-4)
    p, v, d = ev.compute_blocking(bb)
    assert ev.last_plan()["trunk_precision"] == "f16m8"
    idx = [0, 150, 299]
    check((p[idx], v[idx], d[idx]), net.evaluate(bb[idx]), TOL)


def test_load_device_blob_matches_load_memory(nsg):
    """nsg_load_device_blob (the weight blob already in HBM, e.g. after the RCCL broadcast of
    bench.py / dist.broadcast_blob): same network as nsg_load_memory, bit for bit."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime libnsg.so itself is linked against
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    w = nsg.weights.make_random(2, 128, seed=71, bn="random")
    blob = nsg.weights.to_blob(w)
    bb = nsg.synth.random_batch(9, 86, seed=72)
    a = nsg.Evaluator(0, 16, 86, precision="f16x3"); a.load_memory(blob)
    dev = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dev), len(blob)) == 0
    try:
        assert hip.hipMemcpy(dev, blob, len(blob), 1) == 0  # hipMemcpyHostToDevice
        b = nsg.Evaluator(0, 16, 86, precision="f16x3"); b.load_device_blob(dev.value, len(blob))
        for x, y in zip(a.compute_blocking(bb), b.compute_blocking(bb)):
            np.testing.assert_array_equal(x, y)
        with pytest.raises(nsg.NsgError):
            b2 = nsg.Evaluator(0, 16, 86)
            b2.load_device_blob(dev.value, 100)  # truncated blob: refused, no crash
    finally:
        hip.hipFree(dev)


@pytest.mark.parametrize("mx", ["f16m8", "f16m6"])
@pytest.mark.parametrize("channels,batch,ksplit", [(256, 70, 2), (256, 101, 2), (256, 128, 2), (128, 200, 2),
                                                   (256, 1, 4), (256, 7, 4), (256, 19, 4), (256, 32, 4), (256, 40, 4), (256, 64, 4)])
def test_f16m8_k_split_tiles(nsg, oracle, monkeypatch, channels, batch, ksplit, mx):
    """One-board kF16m8 tiles split by K, the whole board resident in eight LDS image buffers.
    Mid batches (one workgroup per board and 128 output channels fills more than half the CUs): the
    two waves of a channel group each run half of the input-channel chunk pairs over all six row
    fragments.  Small batches (at most CUs/4 boards, 256 channels): four workgroups per board, one
    64-channel group each, its four waves one chunk pair apiece (the stem, with two pairs, runs the
    two-halves kernel).  The waves add their accumulators through LDS.  Against the oracle, against
    the f16x3 evaluator on every board, and -- two halves -- against the row-split plan of the same
    arithmetic (the f32 sums differ in their last bit by summation order; the fp8 rounding of the low
    term then turns some of those into differences of the size of the format's own error, ~6e-5 here)."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    wg = batch * (channels // 128)
    if (ksplit == 2 and not (wg <= cus < 2 * wg)) or (ksplit in (3, 4) and batch * ksplit > cus):
        pytest.skip("batch range of this plan depends on the CU count")
    bmax = max(batch, 2)
    if batch <= 8:  # (round 3: up to eight boards run the team trunk by default -- test_team_trunk_small_batches;
        monkeypatch.setenv("NSG_TEAM_TRUNK", "0")  # these per-layer plans are what it falls back to)
    ev, blob = make(nsg, 3, channels, bmax, precision=mx, seed=63)
    bb = nsg.synth.random_batch(batch, 86, seed=64, garbage=True)
    p, v, d = ev.compute_blocking(bb)
    plan = ev.last_plan()
    # the smallest batches split the rows of a four-way K split over two workgroups as well (eight per board)
    rows8 = ksplit == 4 and batch * 8 <= cus
    assert plan["trunk_precision"] == mx and plan["boards_per_group"] == 1 and plan["k_split"] == ksplit
    # (an f16m6 evaluator takes two row groups where three would fit: twelve workgroups per board have no cooperative form)
    three = batch * 12 <= cus and not (mx == "f16m6" and ev.last_launch_kind()[0] == "coop")
    assert plan["row_split"] == (6 if batch * 24 <= cus else 3 if three else 2 if rows8 else 1)
    idx = sorted({0, batch // 2, batch - 1})
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), TOL)
    x3, _ = make(nsg, 3, channels, bmax, precision="f16x3", seed=63)
    p3, v3, d3 = x3.compute_blocking(bb)
    assert float(np.abs(p - p3).max()) < TOL and float(np.abs(v - v3).max()) < TOL and float(np.abs(d - d3).max()) < TOL
    if ksplit == 2:
        monkeypatch.setenv("NSG_CONV_MSPLIT", "2")
        rows, _ = make(nsg, 3, channels, bmax, precision=mx, seed=63)
        pr, vr, dr = rows.compute_blocking(bb)
        plan = rows.last_plan()
        assert plan["row_split"] == 2 and plan["k_split"] == 1
        assert float(np.abs(p - pr).max()) < 3e-4 and float(np.abs(v - vr).max()) < 1e-4 and float(np.abs(d - dr).max()) < 1e-4
    if rows8:  # row fragments are independent: four workgroups per board give the same bits as eight
        monkeypatch.setenv("NSG_ROWSPLIT8_MAX_BATCH", "0")
        four, _ = make(nsg, 3, channels, bmax, precision=mx, seed=63)
        p4, v4, d4 = four.compute_blocking(bb)
        plan = four.last_plan()
        assert plan["row_split"] == 1 and plan["k_split"] == 4
        np.testing.assert_array_equal(p4, p)
        np.testing.assert_array_equal(v4, v)
        np.testing.assert_array_equal(d4, d)
    # a forward is deterministic: the same bits again, also at another slot of the batch
    p2, v2, d2 = ev.compute_blocking(bb[::-1].copy())
    np.testing.assert_array_equal(p2[::-1], p)
    np.testing.assert_array_equal(v2[::-1], v)


def test_f16m8_window_guard(nsg, oracle, monkeypatch):
    """The load-time guard of F16M8: a network whose estimated activations leave the window of the
    fixed-scale e4m3 copies (here BN gamma 512 in the stem: values in the thousands) runs on the
    evaluator's F16X3 copy of the trunk and keeps 1e-3; an ordinary network stays on the MX path."""
    monkeypatch.setenv("NSG_CONV_NFRAG", "4")
    w = nsg.weights.make_random(2, 64, seed=22, bn="random")
    ev = nsg.Evaluator(0, 8, 86, precision="f16m8")
    ev.load_memory(nsg.weights.to_blob(w))
    info = ev.info()
    assert info["f16m8_window_fallback"] == 0 and 1.0 < info["activation_bound_estimate"] < 224.0
    ev.compute_blocking(nsg.synth.random_batch(8, 86, seed=2))
    assert ev.last_plan()["trunk_precision"] == "f16m8"
    w["stem_bn"][0] *= 512.0
    blob = nsg.weights.to_blob(w)
    ev = nsg.Evaluator(0, 8, 86, precision="f16m8")
    ev.load_memory(blob)
    info = ev.info()
    assert info["f16m8_window_fallback"] == 1 and info["activation_bound_estimate"] > 224.0
    bb = nsg.synth.random_batch(8, 86, seed=2)
    out = ev.compute_blocking(bb)
    assert ev.last_plan()["trunk_precision"] == "f16x3"
    ref = oracle.net(blob).evaluate(bb)
    scale = max(1.0, float(np.abs(ref[0]).max()))
    assert float(np.abs(out[0] - ref[0]).max()) <= 1e-4 * scale
    # f16m6 needs no guard
    ev6 = nsg.Evaluator(0, 8, 86, precision="f16m6")
    ev6.load_memory(blob)
    assert ev6.info()["f16m8_window_fallback"] == 0
    out6 = ev6.compute_blocking(bb)
    assert ev6.last_plan()["trunk_precision"] == "f16m6"
    assert float(np.abs(out6[0] - ref[0]).max()) <= 1e-3 * scale


def test_f16m8_extreme_magnitudes_stay_finite(nsg, oracle, monkeypatch):
    """With the guard switched off (NSG_M8_GUARD=0): activations far beyond the fp8 range of the
    correction operands (BN gamma 512 in the stem: values in the thousands) and weights spanning
    many binades: the fixed-scale fp8 copies saturate instead of overflowing, so outputs stay finite
    and degrade no further than plain f16 accuracy (the f16 main term is unaffected)."""
    monkeypatch.setenv("NSG_CONV_NFRAG", "4")
    monkeypatch.setenv("NSG_M8_GUARD", "0")
    w = nsg.weights.make_random(2, 64, seed=22, bn="random")
    rng = np.random.default_rng(4)
    w["b0_w1"] = (w["b0_w1"] * np.exp2(rng.integers(-10, 3, size=w["b0_w1"].shape[:1])).reshape(-1, 1, 1, 1)).astype(np.float32)
    w["stem_bn"][0] *= 512.0
    blob = nsg.weights.to_blob(w)
    ev = nsg.Evaluator(0, 8, 86, precision="f16m8")
    ev.load_memory(blob)
    bb = nsg.synth.random_batch(8, 86, seed=2)
    out = ev.compute_blocking(bb)
    assert ev.last_plan()["trunk_precision"] == "f16m8"
    ref = oracle.net(blob).evaluate(bb)
    assert np.isfinite(out[0]).all() and np.isfinite(out[1]).all() and np.isfinite(out[2]).all()
    scale = max(1.0, float(np.abs(ref[0]).max()))
    assert float(np.abs(out[0] - ref[0]).max()) <= 2e-2 * scale  # the plain-f16 tolerance of this suite
    assert float(np.abs(out[1] - ref[1]).max()) <= 2e-2


def _e2m3_codes(x):
    """OCP MX e2m3 (1 sign, 2 exponent, 3 mantissa bits; bias 1; no inf/nan): nearest code, ties to
    the even code, saturating at 7.5 -- for values already divided by their block scale."""
    mags = np.array([(c / 8.0) if c < 8 else (1 + (c & 7) / 8.0) * 2.0 ** ((c >> 3) - 1) for c in range(32)])
    a = np.abs(x).astype(np.float64)
    hi = np.searchsorted(mags, a, side="left").clip(0, 31)
    lo = (hi - 1).clip(0, 31)
    dl, dh = a - mags[lo], mags[hi] - a
    pick_hi = (dh < dl) | ((dh == dl) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo)
    code = np.where(a >= 7.5, 31, code)
    return (code | ((np.signbit(x)).astype(np.int64) << 5)).astype(np.uint8)


def test_f16m6_trunk_input_encoding(nsg, oracle):
    """The plane expansion of an f16m6 evaluator, byte for byte: per (square, 32-channel chunk) a
    128-byte row [32 x f16 hi][24 B e2m3(hi) + E8M0 exponent, stored in both trailing dwords][24 B e2m3(lo) + exponent], value j of a
    block at bits [6j, 6j+6), exponent = that of the block maximum minus 2 (zero block: 2^-17, codes
    0).  Inputs: real plane bits plus scalar planes with values all over [0, 1]."""
    B = 70  # an MX plan (two-board tiles need > CUs/4 boards)
    ev, _ = make(nsg, 1, 256, B, precision="f16m6", seed=3)
    bb = nsg.synth.random_batch(B, 86, seed=19, garbage=True)
    ev.upload_features(bb)
    ev.forward_resident(B)
    assert ev.last_plan()["trunk_precision"] == "f16m6"
    raw = ev.download_planes_raw(B)
    assert raw.shape == (B, 81, 4 * 128)
    planes = oracle.extract_bits(bb)                                     # [B][86][81] f32
    x = np.zeros((B, 81, 128), dtype=np.float32)
    x[:, :, :86] = np.clip(planes.reshape(B, 86, 81).transpose(0, 2, 1), -65000, 65000)
    rows = raw.reshape(B, 81, 4, 128)
    hi = rows[..., :64].copy().view(np.float16).reshape(B, 81, 4, 32)
    want_hi = x.reshape(B, 81, 4, 32).astype(np.float16)
    np.testing.assert_array_equal(hi.view(np.uint16), want_hi.view(np.uint16))
    want_lo = (x.reshape(B, 81, 4, 32) - want_hi.astype(np.float32)).astype(np.float16)
    for name, off, vals in (("hi", 64, want_hi), ("lo", 96, want_lo)):
        blk = rows[..., off:off + 32]
        e8 = blk[..., 24].astype(np.int64)
        assert (blk[..., 25:28] == 0).all() and (blk[..., 28] == blk[..., 24]).all() and (blk[..., 29:32] == 0).all()
        mx = np.abs(vals.astype(np.float64)).max(axis=-1)
        bits = np.abs(vals).max(axis=-1).astype(np.float16).view(np.uint16).astype(np.int64)
        np.testing.assert_array_equal(e8, (bits >> 10) + 110, err_msg=name)
        codes = np.zeros(blk.shape[:-1] + (32,), dtype=np.uint8)
        words = blk[..., :24].copy().view("<u4").astype(np.uint64).reshape(blk.shape[:-1] + (6,))
        for j in range(32):
            w, sh = divmod(6 * j, 32)
            v = words[..., w] >> np.uint64(sh)
            if sh > 26:
                v = v | (words[..., w + 1] << np.uint64(32 - sh))
            codes[..., j] = (v & np.uint64(63)).astype(np.uint8)
        scaled = vals.astype(np.float64) / np.exp2(e8 - 127.0)[..., None]
        want = _e2m3_codes(scaled)
        same = (codes == want) | ((codes & 31 == 0) & (want & 31 == 0))    # +0 / -0
        assert same.all(), (name, int((~same).sum()), codes[~same][:8], want[~same][:8])
        assert mx.max() > 0


def test_f16m6_block_scales_follow_the_data(nsg, oracle, monkeypatch):
    """The case test_f16m8_extreme_magnitudes_stay_finite relaxes to 2e-2 -- activations in the
    thousands (BN gamma 512 in the stem) and weights spanning thirteen binades -- in f16m6: its
    correction operands carry one exponent per 32 channels instead of fixed scales, so there is no
    clamp window to leave and the error stays at 1e-3 of the output range."""
    monkeypatch.setenv("NSG_CONV_NFRAG", "4")
    w = nsg.weights.make_random(2, 64, seed=22, bn="random")
    rng = np.random.default_rng(4)
    w["b0_w1"] = (w["b0_w1"] * np.exp2(rng.integers(-10, 3, size=w["b0_w1"].shape[:1])).reshape(-1, 1, 1, 1)).astype(np.float32)
    w["stem_bn"][0] *= 512.0
    blob = nsg.weights.to_blob(w)
    ev = nsg.Evaluator(0, 8, 86, precision="f16m6")
    ev.load_memory(blob)
    bb = nsg.synth.random_batch(8, 86, seed=2)
    out = ev.compute_blocking(bb)
    assert ev.last_plan()["trunk_precision"] == "f16m6"
    ref = oracle.net(blob).evaluate(bb)
    scale = max(1.0, float(np.abs(ref[0]).max()))
    assert np.isfinite(out[0]).all()
    assert float(np.abs(out[0] - ref[0]).max()) <= 1e-3 * scale
    assert float(np.abs(out[1] - ref[1]).max()) <= 1e-3 and float(np.abs(out[2] - ref[2]).max()) <= 1e-3


@pytest.mark.parametrize("mx", ["f16m8", "f16m6"])
def test_f16m8_ragged_batch_sizes_against_f16x3(nsg, mx):
    """Every board of ragged batches on both sides of the plan boundaries (one- and two-board
    tiles, row-split one-board tiles, one and two chains, the f16x3 fallback) against the
    f32-equivalent f16x3 evaluator."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    sizes = [cus // 4, cus // 4 + 1, cus // 2 - 1, cus // 2, cus // 2 + 1, 3 * cus // 4 - 1, 3 * cus // 4 + 1, cus, cus + 1,
             3 * cus // 2 - 1, 2 * cus - 1, 2 * cus + 1, 2 * cus + 88]
    bmax = max(sizes)
    w = nsg.weights.make_random(1, 256, seed=50, bn="random")
    blob = nsg.weights.to_blob(w)
    a = nsg.Evaluator(0, bmax, 86, precision=mx); a.load_memory(blob)
    b = nsg.Evaluator(0, bmax, 86, precision="f16x3"); b.load_memory(blob)
    bb = nsg.synth.random_batch(bmax, 86, seed=51)
    seen = set()
    for n in sizes:
        pa, va, da = a.compute_blocking(bb[:n])
        plan = a.last_plan()
        seen.add((plan["trunk_precision"], plan["boards_per_group"], plan["chains"]))
        pb, vb, db = b.compute_blocking(bb[:n])
        assert np.isfinite(pa).all()
        assert float(np.abs(pa - pb).max()) < TOL, (n, plan)
        assert float(np.abs(va - vb).max()) < TOL and float(np.abs(da - db).max()) < TOL
    assert (mx, 1, 1) in seen and (mx, 2, 1) in seen and (mx, 2, 2) in seen, seen


@pytest.mark.parametrize("precision", ["f16m8", "f16x3"])
def test_staggered_chains_bit_identical(nsg, monkeypatch, precision):
    """NSG_CHAIN_DELAY_US=-1 (off by default): with all tiles resident at once on more than half the
    CUs (batch = 2 x CUs) the forward runs as two chains, the second started a measured fraction of
    a layer later.  The first forward measures the layer time unstaggered, the following ones are
    staggered; every one is bit-identical to a single chain."""
    monkeypatch.setenv("NSG_CHAIN_DELAY_US", "-1")
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    batch = 2 * cus
    bb = nsg.synth.random_batch(batch, 86, seed=91)
    ev, _ = make(nsg, 2, 256, batch, precision=precision, seed=35)
    outs = [ev.compute_blocking(bb) for _ in range(3)]
    assert ev.last_plan()["chains"] == 2
    monkeypatch.setenv("NSG_CHAINS", "1")
    ev1, _ = make(nsg, 2, 256, batch, precision=precision, seed=35)
    ref = ev1.compute_blocking(bb)
    assert ev1.last_plan()["chains"] == 1
    for o in outs:
        for x, y in zip(o, ref):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("mx", ["f16m8", "f16m6"])
def test_f16m8_persistent_trunk_kernel_bit_identical(nsg, monkeypatch, mx):
    """NSG_TRUNK_KERNEL=1: all 3x3 layers in one launch (a workgroup owns its boards through every
    layer, no grid barrier).  Same arithmetic, so bit-identical to per-layer launches."""
    bb = nsg.synth.random_batch(300, 86, seed=79)
    monkeypatch.setenv("NSG_SPLIT_BATCH", "0")  # one plan for the whole batch on both sides (300 = CUs + 44 would run as two parts)
    monkeypatch.setenv("NSG_TRUNK_KERNEL", "0")
    ev, _ = make(nsg, 3, 256, 300, precision=mx, seed=34)
    p, v, d = ev.compute_blocking(bb)
    monkeypatch.setenv("NSG_TRUNK_KERNEL", "1")
    ev1, _ = make(nsg, 3, 256, 300, precision=mx, seed=34)
    p1, v1, d1 = ev1.compute_blocking(bb)
    assert ev1.last_plan()["trunk_precision"] == mx

    np.testing.assert_array_equal(p, p1)
    np.testing.assert_array_equal(v, v1)
    np.testing.assert_array_equal(d, d1)


def test_f16m8_chains_bit_identical(nsg, monkeypatch):
    """More tiles than CUs: two half-batch chains on two streams, bit-identical to one chain."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    batch = 2 * cus + 37
    bb = nsg.synth.random_batch(batch, 86, seed=78)
    monkeypatch.setenv("NSG_SPLIT_BATCH", "0")  # (just above 2 CUs boards the default is a full chip of tiles + the remainder)
    ev, _ = make(nsg, 2, 256, batch, precision="f16m8", seed=33)
    p, v, d = ev.compute_blocking(bb)
    plan = ev.last_plan()
    assert plan["chains"] == 2 and plan["trunk_precision"] == "f16m8"
    monkeypatch.setenv("NSG_CHAINS", "1")
    ev1, _ = make(nsg, 2, 256, batch, precision="f16m8", seed=33)
    p1, v1, d1 = ev1.compute_blocking(bb)
    np.testing.assert_array_equal(p, p1)
    np.testing.assert_array_equal(v, v1)
    np.testing.assert_array_equal(d, d1)


@pytest.mark.parametrize("mx", ["f16m8", "f16m6"])
def test_f16m8_full_size_properties(nsg, oracle, mx):
    """kF16m8 / kF16m6 at BASELINE's full size (20x256, B=512): a sample of boards against the
    oracle at the north_star tolerance, bit-exact independence from batch composition."""
    ev, blob = make(nsg, 20, 256, 512, precision=mx, seed=4, bn="identity")
    bb = nsg.synth.random_batch(512, 86, seed=8)
    p, v, d = ev.compute_blocking(bb)
    idx = [0, 255, 511]
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), TOL)
    perm = np.random.default_rng(1).permutation(512)
    p2, v2, d2 = ev.compute_blocking(bb[perm])
    np.testing.assert_array_equal(p2, p[perm])
    np.testing.assert_array_equal(v2, v[perm])
    assert ((v >= 0) & (v <= 1)).all() and ((d >= 0) & (d <= 1)).all() and np.isfinite(p).all()


def test_f16x3_extreme_magnitudes(nsg, oracle):
    """Weights spanning many binades (the per-tensor power-of-two scale must keep
    hi/lo in f16 range) and large activations (BN gamma 8): still f32-equivalent."""
    w = nsg.weights.make_random(2, 64, seed=21, bn="random")
    rng = np.random.default_rng(3)
    for k in ("b0_w1", "b1_w2"):
        w[k] = (w[k] * np.exp2(rng.integers(-12, 3, size=w[k].shape[:1])).reshape(-1, 1, 1, 1)).astype(np.float32)
    w["stem_bn"][0] *= 8.0
    blob = nsg.weights.to_blob(w)
    ev = nsg.Evaluator(0, 8, 86, precision="f16x3")
    ev.load_memory(blob)
    bb = nsg.synth.random_batch(8, 86, seed=2)
    out = ev.compute_blocking(bb)
    ref = oracle.net(blob).evaluate(bb)
    scale = max(1.0, float(np.abs(ref[0]).max()))
    assert float(np.abs(out[0] - ref[0]).max()) <= 2e-5 * scale
    assert float(np.abs(out[1] - ref[1]).max()) <= 1e-5 and float(np.abs(out[2] - ref[2]).max()) <= 1e-5


def test_nonblocking_await_contract(nsg, oracle):
    """computeNonBlocking -> await; results defined only after await (trt.cc:234-283)."""
    ev, blob = make(nsg, 2, 64, 64, seed=6)
    bb = nsg.synth.random_batch(64, 86, seed=6)
    out = ev.compute_nonblocking(bb)
    ev.await_()
    assert not ev.is_computing()
    check(out, oracle.net(blob).evaluate(bb), 2e-4)
    # caller-owned output buffers are reused across calls (Evaluator owns them)
    pol = np.empty((64, 2187), np.float32); win = np.empty(64, np.float32); drw = np.empty(64, np.float32)
    ev.compute_blocking(bb[:10], policy=pol, win=win, draw=drw)
    check((pol[:10], win[:10], drw[:10]), (out[0][:10], out[1][:10], out[2][:10]), 1e-5)


def test_device_side_legal_move_gather(nsg):
    """the legal-move lookup on the device.  Gathered logits are bit-identical
    to indexing the full 2187-wide rows; the device softmax matches the host's f32 softmax;
    ragged counts including positions with no moves and with the 593-move maximum."""
    ev, _ = make(nsg, 2, 64, 40, seed=9)
    bb = nsg.synth.random_batch(40, 86, seed=9)
    p, v, d = ev.compute_blocking(bb)
    rng = np.random.default_rng(5)
    counts = rng.integers(1, 200, size=40)
    counts[3] = 0
    counts[7] = 593
    counts[39] = 1
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    idx = np.concatenate([rng.choice(2187, size=c, replace=False) for c in counts]).astype(np.uint16)
    g, v2, d2 = ev.compute_gather_blocking(bb, idx, off)
    want = np.concatenate([p[b, idx[off[b]:off[b + 1]]] for b in range(40)])
    np.testing.assert_array_equal(g, want)
    np.testing.assert_array_equal(v2, v)
    np.testing.assert_array_equal(d2, d)
    s, _, _ = ev.compute_gather_blocking(bb, idx, off, softmax=True)
    for b in range(40):
        row = want[off[b]:off[b + 1]].astype(np.float32)
        if row.size == 0:
            continue
        e = np.exp(row - row.max(), dtype=np.float32)
        ref = e / e.sum(dtype=np.float32)
        np.testing.assert_allclose(s[off[b]:off[b + 1]], ref, rtol=2e-6, atol=1e-7)
        assert abs(float(s[off[b]:off[b + 1]].sum()) - 1.0) < 1e-5
    bad = off.copy(); bad[5] = bad[6] + 1
    with pytest.raises(nsg.NsgError):
        ev.compute_gather_blocking(bb, idx, bad)


def test_resident_path_equals_host_path(nsg):
    ev, _ = make(nsg, 2, 64, 16, seed=2)
    bb = nsg.synth.random_batch(16, 86, seed=2)
    a = ev.compute_blocking(bb)
    ev.upload_features(bb)
    ev.forward_resident(16)
    b = ev.download_outputs(16)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_error_behaviour(nsg, tmp_path):
    ev = nsg.Evaluator(0, 4, 86)
    bb = nsg.synth.random_batch(4, 86)
    with pytest.raises(nsg.NsgError):  # compute before load
        ev.compute_blocking(bb)
    with pytest.raises(nsg.NsgError):  # trt.cc:34-36: "Could not open the file"
        ev.load(str(tmp_path / "missing.nsgw"))
    bad = tmp_path / "bad.nsgw"
    bad.write_bytes(b"not a weight file")
    with pytest.raises(nsg.NsgError):
        ev.load(str(bad))
    w = nsg.weights.make_random(1, 64, in_channels=93)  # wrong plane count
    with pytest.raises(nsg.NsgError):
        ev.load_memory(nsg.weights.to_blob(w))
    ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(1, 64)))
    with pytest.raises(nsg.NsgError):  # BatchSize > BatchSizeMax (trt.cc:237 assert)
        ev.compute_blocking(nsg.synth.random_batch(5, 86))
    path = tmp_path / "ok.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(1, 64, seed=3))
    ev2 = nsg.Evaluator(0, 4, 86)
    ev2.load(str(path))
    assert ev2.info()["loaded"] == 1 and ev2.info()["channels"] == 64


@pytest.mark.parametrize("precision,tol", [("f16x3", 1e-4), ("f16m8", 1e-3), ("f16m6", 1e-3), ("fp16", 2e-2), ("bf16", 1.5e-1)])
def test_reduced_precision_paths(nsg, oracle, precision, tol):
    """16-bit operand paths (f32 accumulate).  f16x3 (split hi/lo, three MFMAs per
    MAC) is f32-equivalent and is held to 1e-4; plain f16/bf16 are looser than the
    north_star's 1e-3 and their tolerance is stated here; DESIGN.md discusses it."""
    ev, blob = make(nsg, 4, 128, 16, precision=precision, seed=5, bn="identity")
    bb = nsg.synth.random_batch(12, 86, seed=5)
    out = ev.compute_blocking(bb)
    ref = oracle.net(blob).evaluate(bb)
    check(out, ref, tol)
    assert np.isfinite(out[0]).all()


@pytest.mark.parametrize("env,expect", [
    ({"NSG_TRUNK_KERNEL": "1"}, {}),
    ({"NSG_CONV_NB": "1"}, {"boards_per_group": 1}),
    ({"NSG_CONV_NFRAG": "2"}, {"fragments_per_wave": 2}),
    ({"NSG_CONV_NFRAG": "1", "NSG_CONV_NB": "2"}, {"fragments_per_wave": 1, "boards_per_group": 2}),
    ({"NSG_CONV_NFRAG": "4", "NSG_CONV_NB": "2", "NSG_CONV_NWAVES": "2"},
     {"fragments_per_wave": 4, "boards_per_group": 2, "waves_per_group": 2}),
])
@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_alternative_launch_plans_agree(nsg, oracle, monkeypatch, env, expect, precision):
    """The tuning knobs (persistent one-launch trunk, tile shapes) change the launch
    plan, not the arithmetic: results stay within the oracle tolerance."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ev, blob = make(nsg, 2, 256, 70, precision=precision, seed=31)
    bb = nsg.synth.random_batch(70, 86, seed=31)
    p, v, d = ev.compute_blocking(bb)
    plan = ev.last_plan()
    for k, want in expect.items():
        assert plan[k] == want, (plan, env)
    idx = [0, 35, 69]
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), 2e-4)


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_half_batch_chains_bit_identical(nsg, oracle, monkeypatch, precision):
    """A batch with more 2-board tiles than CUs runs as two half-batch chains on two
    streams (odd half sizes included); the split changes scheduling only, so outputs are
    bit-identical to the single-chain run and match the oracle."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    batch = 2 * cus + 71
    bb = nsg.synth.random_batch(batch, 86, seed=77)
    ev, blob = make(nsg, 2, 256, batch, precision=precision, seed=32)
    p, v, d = ev.compute_blocking(bb)
    assert ev.last_plan()["chains"] == 2
    monkeypatch.setenv("NSG_CHAINS", "1")
    ev1, _ = make(nsg, 2, 256, batch, precision=precision, seed=32)
    p1, v1, d1 = ev1.compute_blocking(bb)
    assert ev1.last_plan()["chains"] == 1
    np.testing.assert_array_equal(p, p1)
    np.testing.assert_array_equal(v, v1)
    np.testing.assert_array_equal(d, d1)
    idx = [0, batch // 2 - 1, batch // 2, batch // 2 + 1, batch - 1]
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), 2e-4)


def test_host_stats_and_profiler_markers(nsg, tmp_path):
    """nsg_get_stats (average batch = positions / batches, mcts::Statistics) and the roctx ranges:
    with NSG_ROCTX=1 the marker library is resolved at first use and the outputs do not change."""
    import subprocess
    import sys
    ev, _ = make(nsg, 2, 64, 16, seed=2)
    bb = nsg.synth.random_batch(16, 86, seed=2)
    assert ev.stats() == {"batches": 0, "positions": 0, "average_batch": 0.0}
    ref = ev.compute_blocking(bb)
    ev.compute_blocking(bb[:4])
    ev.compute_blocking(bb[:1])
    assert ev.stats() == {"batches": 3, "positions": 21, "average_batch": 7.0}
    np.save(tmp_path / "ref.npy", ref[0])
    code = ("import importlib, numpy as np, sys; nsg = importlib.import_module('nshogi-engine_amd');"
            "ev = nsg.Evaluator(0, 16, 86); ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(2, 64, seed=2, bn='random')));"
            "p, v, d = ev.compute_blocking(nsg.synth.random_batch(16, 86, seed=2));"
            f"assert np.array_equal(p, np.load(r'{tmp_path / 'ref.npy'}')); print('markers ok')")
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root,
                       env=dict(__import__("os").environ, NSG_ROCTX="1"))
    assert r.returncode == 0 and "markers ok" in r.stdout, r.stdout + r.stderr


def test_tuning_variables_are_validated(nsg, monkeypatch):
    """A tuning variable outside its domain is refused when the evaluator is created."""
    for name, bad in (("NSG_CONV_NB", "3"), ("NSG_CONV_NFRAG", "3"), ("NSG_CHAINS", "two"), ("NSG_TRUNK_KERNEL", "yes"),
                      ("NSG_CONV_MSPLIT", "0"), ("NSG_TEAM_MEMBERS", "64"), ("NSG_TEAM_MAX_BATCH", "17")):
        monkeypatch.setenv(name, bad)
        with pytest.raises(nsg.NsgError, match=name):
            nsg.Evaluator(0, 4, 86)
        monkeypatch.delenv(name)
    monkeypatch.setenv("NSG_CONV_NB", "2")
    nsg.Evaluator(0, 4, 86).close()


@pytest.mark.parametrize("precision", ["fp32", "f16m6"])
def test_load_shared_same_device(nsg, precision):
    """nsg_load_shared: a second evaluator adopts the first one's network without reading the model
    again (same device: one shared copy of the packed weights).  Same outputs bit for bit, different
    batch capacities, and the source may be destroyed first."""
    blob = nsg.weights.to_blob(nsg.weights.make_random(2, 128, seed=81, bn="random"))
    bb = nsg.synth.random_batch(40, 86, seed=82)
    a = nsg.Evaluator(0, 40, 86, precision=precision)
    a.load_memory(blob)
    ref = a.compute_blocking(bb)
    b = nsg.Evaluator(0, 64, 86, precision=precision)
    b.load_shared(a)
    assert b.info()["loaded"] == 1 and b.info()["channels"] == 128 and b.info()["param_count"] == a.info()["param_count"]
    for x, y in zip(b.compute_blocking(bb), ref):
        np.testing.assert_array_equal(x, y)
    a.close()  # the shared weights stay alive with their last user
    for x, y in zip(b.compute_blocking(bb), ref):
        np.testing.assert_array_equal(x, y)
    c = nsg.Evaluator(0, 8, 86, precision="f16x3" if precision == "fp32" else "fp32")
    with pytest.raises(nsg.NsgError, match="precision"):
        c.load_shared(b)
    d = nsg.Evaluator(0, 8, 86, precision=precision)
    with pytest.raises(nsg.NsgError):
        d.load_shared(c)  # nothing loaded in the source


def test_compute_from_a_thread_that_never_bound_the_device(nsg):
    """Every compute entry binds the evaluator's device itself (the LDS-size attribute of
    the tile kernels was set for the CALLING thread's current device).  A fresh thread that never
    called resetGPU drives the evaluator; with a second GPU present, an evaluator on GPU 1 that took
    its network from GPU 0 by a peer copy (nsg_load_shared) is driven the same way and must return
    GPU 0's outputs bit for bit."""
    import threading
    import torch
    blob = nsg.weights.to_blob(nsg.weights.make_random(2, 128, seed=91, bn="random"))
    bb = nsg.synth.random_batch(70, 86, seed=92)
    a = nsg.Evaluator(0, 70, 86, precision="f16m6")
    a.load_memory(blob)
    ref = a.compute_blocking(bb)
    evs = [a]
    if torch.cuda.device_count() >= 2:
        b = nsg.Evaluator(1, 70, 86, precision="f16m6")
        b.load_shared(a)
        evs.append(b)
    got, err = {}, []

    def drive(i, ev):
        try:
            got[i] = ev.compute_blocking(bb)          # >64 KiB of LDS per workgroup at this size
            ev.upload_features(bb)
            ev.forward_resident(70)
            got[(i, "resident")] = ev.download_outputs(70)
        except Exception as e:  # noqa: BLE001
            err.append(repr(e))

    ts = [threading.Thread(target=drive, args=(i, ev)) for i, ev in enumerate(evs)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not err, err
    for i in range(len(evs)):
        for x, y in zip(got[i], ref):
            np.testing.assert_array_equal(x, y)
        for x, y in zip(got[(i, "resident")], ref):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("mx", ["f16m8", "f16m6"])
def test_two_part_batches(nsg, oracle, monkeypatch, mx):
    """Between the batch sizes whose plans fill the chip exactly, a batch runs as a full part with the
    plan of the size below plus the remainder with its own plan, on two streams (CUs/2 + r boards:
    two-way K split + the small-batch plan of r; CUs + r: one-board tiles + r).  Against the oracle,
    against the f16x3 evaluator on every board, and against the one-plan run (NSG_SPLIT_BATCH=0:
    other summation orders, same arithmetic)."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    sizes = [cus // 2 + 1, cus // 2 + 23, cus // 2 + 3 * cus // 16, cus + 5, cus + cus // 4, cus + cus // 2 - 1, 2 * cus + 9]
    bmax = 2 * cus + 9
    ev, blob = make(nsg, 2, 256, bmax, precision=mx, seed=71)
    x3, _ = make(nsg, 2, 256, bmax, precision="f16x3", seed=71)
    monkeypatch.setenv("NSG_SPLIT_BATCH", "0")
    one, _ = make(nsg, 2, 256, bmax, precision=mx, seed=71)
    bb = nsg.synth.random_batch(bmax, 86, seed=72, garbage=True)
    net = oracle.net(blob)
    for n in sizes:
        p, v, d = ev.compute_blocking(bb[:n])
        # (round 4: where every one-board tile of the batch is resident at once, 9/16 CUs .. CUs boards, an f16m6
        # evaluator runs the persistent trunk launch instead -- one chain; the same bits as its per-layer kernels;
        # likewise 3/4 CUs .. CUs two-board tiles)
        t2 = (n + 1) // 2  # two-board tiles
        persistent = mx == "f16m6" and (9 * cus <= 16 * n <= 16 * cus or
                                        (ev.last_plan()["boards_per_group"] == 2 and 3 * cus <= 4 * t2 and t2 <= cus))
        assert ev.last_plan()["chains"] == (1 if persistent else 2) and ev.last_plan()["trunk_precision"] == mx, (n, ev.last_plan())
        p1, v1, d1 = one.compute_blocking(bb[:n])
        assert one.last_plan()["chains"] == (1 if n <= 2 * cus else 2)  # (more tiles than CUs: two half-batch chains)
        p3, v3, d3 = x3.compute_blocking(bb[:n])
        assert float(np.abs(p - p3).max()) < TOL and float(np.abs(v - v3).max()) < TOL and float(np.abs(d - d3).max()) < TOL
        assert float(np.abs(p - p1).max()) < 5e-4 and float(np.abs(v - v1).max()) < 2e-4
        first = cus // 2 if n <= cus else cus if n <= 2 * cus else 2 * cus
        idx = [0, first - 1, first, n - 1]
        check((p[idx], v[idx], d[idx]), net.evaluate(bb[idx]), TOL)
    # one board more than the ranges: a single plan again
    ev.compute_blocking(bb[: cus // 2 + 3 * cus // 16 + 1])
    assert ev.last_plan()["chains"] == 1
    ev.compute_blocking(bb[: cus + cus // 2])
    assert ev.last_plan()["chains"] == 1


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("f16x3", 2e-4), ("f16m6", TOL), ("f16m8", TOL)])
def test_custom_features_v1_93_planes(nsg, oracle, monkeypatch, precision, tol):
    """`evaluate::preset::CustomFeaturesV1` (/root/reference/src/evaluate/preset.h:68-122) has 93 planes, not
    86: the plane count is the evaluator's NumChannels argument (trt.h:44) and the stem's input width, nothing
    else depends on it.  Both a small-tile and a full-tile batch, outputs and trunk against the oracle; a
    weight file for another plane count is refused by name."""
    w = nsg.weights.make_random(2, 256, in_channels=93, seed=93, bn="random")
    blob = nsg.weights.to_blob(w)
    net = oracle.net(blob)
    for batch, full in ((5, False), (6, True)):
        if full:
            monkeypatch.setenv("NSG_CONV_NFRAG", "4")
        ev = nsg.Evaluator(0, 8, 93, precision=precision)
        ev.load_memory(blob)
        assert ev.info()["num_channels"] == 93
        bb = nsg.synth.random_batch(batch, 93, seed=7 + batch, garbage=True)
        check(ev.compute_blocking(bb), net.evaluate(bb), tol)
        ev.close()
    ev = nsg.Evaluator(0, 8, 86, precision=precision)
    with pytest.raises(nsg.NsgError, match="93 input planes"):
        ev.load_memory(blob)
    ev.close()


@pytest.mark.parametrize("mx", ["f16m8", "f16m6"])
@pytest.mark.parametrize("channels,batch,ss", [(256, 65, 4), (256, 101, 4), (256, 128, 4), (256, 129, 2), (256, 200, 2),
                                               (256, 256, 2), (192, 100, 4), (384, 70, 4), (128, 250, 4)])
def test_slab_split_tiles(nsg, oracle, monkeypatch, channels, batch, ss, mx):
    """Mid batches in the MX arithmetic: two-board tiles of one (or two) 64-channel groups per workgroup whose
    waves split every chunk pair's SLABS between them -- each wave runs its own static subset of the 27 slabs
    for all eleven edge-packed row fragments, the group adds its accumulators up through LDS in two rounds.
    Against the oracle, against the f16x3 evaluator on EVERY board, and against the plans these tiles replace
    (NSG_SLAB_SPLIT=0: one-board tiles, K split): the f32 sums differ by summation order only."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    wgs = ((batch + 1) // 2) * (channels // (256 // ss))
    if not (wgs <= cus < 2 * wgs) or (ss == 2 and ((batch + 1) // 2) * (channels // 64) <= cus) or \
            (channels == 256 and batch * 4 <= cus):
        pytest.skip("batch range of this plan depends on the CU count")
    monkeypatch.setenv("NSG_SPLIT_BATCH", "0")  # one plan for the whole batch (129 would run as 128 + 1)
    monkeypatch.setenv("NSG_SLAB_SPLIT", "1")   # opt-in: measured slower than the plans it would replace
    ev, blob = make(nsg, 3, channels, batch, precision=mx, seed=163)
    bb = nsg.synth.random_batch(batch, 86, seed=164, garbage=True)
    p, v, d = ev.compute_blocking(bb)
    plan = ev.last_plan()
    assert plan["trunk_precision"] == mx and plan["boards_per_group"] == 2 and plan["slab_split"] == ss, plan
    idx = sorted({0, 1, batch // 2, batch - 2, batch - 1})
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), TOL)
    x3, _ = make(nsg, 3, channels, batch, precision="f16x3", seed=163)
    p3, v3, d3 = x3.compute_blocking(bb)
    assert float(np.abs(p - p3).max()) < TOL and float(np.abs(v - v3).max()) < TOL and float(np.abs(d - d3).max()) < TOL
    monkeypatch.setenv("NSG_SLAB_SPLIT", "0")
    old, _ = make(nsg, 3, channels, batch, precision=mx, seed=163)
    po, vo, do = old.compute_blocking(bb)
    assert old.last_plan()["slab_split"] == 1
    assert float(np.abs(p - po).max()) < 3e-4 and float(np.abs(v - vo).max()) < 1e-4 and float(np.abs(d - do).max()) < 1e-4
    # slot independence: the same boards in another order give the same bits
    perm = np.random.default_rng(5).permutation(batch)
    p2, v2, d2 = ev.compute_blocking(bb[perm])  # (the tuning variables were read when `ev` was created)
    np.testing.assert_array_equal(p2, p[perm])
    np.testing.assert_array_equal(v2, v[perm])


@pytest.mark.parametrize("precision", ["f16m6", "f16x3"])
@pytest.mark.parametrize("blocks", [2, 5])
def test_team_trunk_small_batches(nsg, oracle, monkeypatch, precision, blocks):
    """Batches of up to sixteen boards: every 3x3 layer in ONE persistent launch, a board per team of 96 / 48 / 32 / 16
    workgroups that hand their 16-channel output slices to each other through agent-scope stores / loads, the payload
    being its own flag (kernels/team_trunk.hip), in the kF16x3 arithmetic.  Against the oracle, against the per-layer
    kernels of the same arithmetic (NSG_TEAM_TRUNK=0: only the f32 summation order differs), bit-identical whatever
    else is in the batch and from launch to launch (the hand-off images alternate and are restored between launches
    of different sizes), for batch sizes 1..8 and 9..16."""
    ev, blob = make(nsg, blocks, 256, 8, precision=precision, seed=300 + blocks)
    net = oracle.net(blob)
    bb = nsg.synth.random_batch(8, 86, seed=301, garbage=True)
    ref = net.evaluate(bb)
    outs = {}
    for n in (8, 1, 3, 5, 2, 8):
        p, v, d = ev.compute_blocking(bb[:n])
        plan = ev.last_plan()
        assert plan["trunk_precision"] == "f16x3" and plan["waves_per_group"] == 8 and plan["k_split"] == 8, plan
        check((p, v, d), tuple(r[:n] for r in ref), 2e-4)
        if n in outs:  # the same launch again, after others in between: the same bits
            np.testing.assert_array_equal(p, outs[n][0])
        outs[n] = (p, v, d)
    for n in (1, 3, 5):  # a board's outputs do not depend on how many other boards share the launch
        np.testing.assert_array_equal(outs[n][0], outs[8][0][:n])
        np.testing.assert_array_equal(outs[n][1], outs[8][1][:n])
    # a board alone in slot 0 == the same board in slot 6 of a full batch
    p6, v6, _ = ev.compute_blocking(bb[6:7])
    np.testing.assert_array_equal(p6[0], outs[8][0][6])
    # device-resident forwards back to back (no await between them)
    ev.upload_features(bb)
    for _ in range(5):
        ev.forward_resident(8)
    pr, vr, dr = ev.download_outputs(8)
    np.testing.assert_array_equal(pr, outs[8][0])
    # the per-layer kernels of the same arithmetic
    monkeypatch.setenv("NSG_TEAM_TRUNK", "0")
    old, _ = make(nsg, blocks, 256, 8, precision="f16x3", seed=300 + blocks)
    po, vo, do = old.compute_blocking(bb)
    assert old.last_plan()["waves_per_group"] != 8
    assert float(np.abs(po - outs[8][0]).max()) < 1e-4 and float(np.abs(vo - outs[8][1]).max()) < 1e-4
    # nine to sixteen boards: 16 workgroups per board, the whole board each; seventeen: not a team batch
    monkeypatch.delenv("NSG_TEAM_TRUNK")
    big, _ = make(nsg, blocks, 256, 17, precision=precision, seed=300 + blocks)
    bb17 = nsg.synth.random_batch(17, 86, seed=302, garbage=True)
    ref17 = net.evaluate(bb17)
    p16, v16, d16 = big.compute_blocking(bb17[:16])
    plan = big.last_plan()
    assert plan["waves_per_group"] == 8 and plan["k_split"] == 8 and plan["row_split"] == 1, plan
    check((p16, v16, d16), tuple(r[:16] for r in ref17), 2e-4)
    p9, v9, d9 = big.compute_blocking(bb17[:9])
    np.testing.assert_array_equal(p9, p16[:9])
    p8, _, _ = big.compute_blocking(bb17[:8])  # 32 workgroups per board
    assert float(np.abs(p8 - p16[:8]).max()) < 1e-4
    big.compute_blocking(bb17)
    assert big.last_plan()["waves_per_group"] != 8


def test_team_trunk_handoff_images_across_launches(nsg):
    """The team trunk's hand-off images are reused within a launch (four rotate) and across launches (two sets
    alternate, a launch restores what the one before left behind): a stale piece taken for a new one, or a piece
    missed, shows as a wrong board.  Sixty launches of changing size over changing selections of sixteen boards, some
    back to back without a host wait: every board's outputs are the bits of its first evaluation, whatever ran before,
    whichever slot it sits in and however many workgroups share its board (96 / 48 / 32 / 16)."""
    ev, _ = make(nsg, 3, 256, 16, precision="f16m6", seed=910)
    bb = nsg.synth.random_batch(16, 86, seed=911, garbage=True)
    p0, v0, d0 = ev.compute_blocking(bb)
    assert ev.last_plan()["waves_per_group"] == 8 and ev.last_plan()["k_split"] == 8  # the team trunk
    rng = np.random.default_rng(912)
    for step in range(60):
        n = int(rng.integers(1, 17))
        pick = rng.permutation(16)[:n]
        if step % 4 == 3:  # device-resident, three forwards queued behind each other
            ev.upload_features(bb[pick])
            for _ in range(3):
                ev.forward_resident(n)
            p, v, d = ev.download_outputs(n)
        else:
            p, v, d = ev.compute_blocking(bb[pick])
        np.testing.assert_array_equal(p, p0[pick], err_msg=f"step {step}: {n} boards {pick.tolist()}")
        np.testing.assert_allclose(v, v0[pick], rtol=0, atol=1e-6)  # (the value MLP's plan follows the batch size)
        np.testing.assert_allclose(d, d0[pick], rtol=0, atol=1e-6)


def test_team_trunk_two_evaluators_share_a_device(nsg, oracle):
    """One team launch per device at a time: an evaluator whose neighbour has a team launch in flight waits for it to
    drain before it launches its own (never two persistent launches holding each other's CUs, and never a
    timing-dependent choice of kernels: both evaluators return the same bits in every round)."""
    a, blob = make(nsg, 3, 256, 8, precision="f16m6", seed=310)
    b = nsg.Evaluator(0, 8, 86, precision="f16m6")
    b.load_memory(blob)
    bb = nsg.synth.random_batch(8, 86, seed=311)
    ref = oracle.net(blob).evaluate(bb)
    pa = [np.empty((8, 2187), np.float32), np.empty(8, np.float32), np.empty(8, np.float32)]
    pb = [np.empty((8, 2187), np.float32), np.empty(8, np.float32), np.empty(8, np.float32)]
    for _ in range(20):
        a.compute_nonblocking(bb, policy=pa[0], win=pa[1], draw=pa[2])
        b.compute_nonblocking(bb, policy=pb[0], win=pb[1], draw=pb[2])
        a.await_()
        b.await_()
        check(tuple(pa), ref, 2e-4)
        np.testing.assert_array_equal(pa[0], pb[0])
        np.testing.assert_array_equal(pa[1], pb[1])
        assert a.last_plan()["waves_per_group"] == 8 and b.last_plan()["waves_per_group"] == 8


def test_planes_download_after_a_team_forward_fails_loudly(nsg):
    """The team trunk decodes the feature bitboards inside its first layer: after such a forward there is no plane
    buffer to read back, and the debug read-back says so instead of returning the planes of an earlier batch."""
    ev, _ = make(nsg, 1, 256, 32, precision="f16m6", seed=4)
    bb = nsg.synth.random_batch(32, 86, seed=20)
    ev.compute_blocking(bb)  # 32 boards: per-layer kernels, planes in memory
    assert ev.download_planes_raw(32).shape[0] == 32
    ev.compute_blocking(bb[:4])
    assert ev.last_plan()["k_split"] == 8  # the team trunk
    with pytest.raises(nsg.NsgError, match="team trunk"):
        ev.download_planes_raw(4)


def test_team_trunk_192_channels(nsg, oracle, monkeypatch):
    """The team trunk on a 192-channel net (BASELINE configs[1]'s width): twelve weight fragments, so teams of 72 / 36 /
    24 / 12 workgroups per board, six of a member's eight waves with a K chunk.  Against the oracle, bit-identical
    across team sizes' shared boards and from launch to launch, against the per-layer f16x3 kernels."""
    ev, blob = make(nsg, 4, 192, 16, precision="f16m6", seed=520)
    bb = nsg.synth.random_batch(16, 86, seed=521, garbage=True)
    ref = oracle.net(blob).evaluate(bb)
    cus = ev.info()["compute_units"]
    outs = {}
    for n in (16, 1, 3, 2, 5, 8, 9, 16, 1):
        p, v, d = ev.compute_blocking(bb[:n])
        plan, ts = ev.last_plan(), ev.team_stats()
        assert plan["trunk_precision"] == "f16x3" and plan["waves_per_group"] == 8 and plan["k_split"] == 8, plan
        want = next(12 * rg for rg in (6, 3, 2, 1) if n * 12 * rg <= cus)
        assert ts["enabled"] == 1 and ts["members_last"] == want and ts["fallbacks"] == 0, (n, ts)
        check((p, v, d), tuple(r[:n] for r in ref), 2e-4)
        if n in outs:
            np.testing.assert_array_equal(p, outs[n][0])
            np.testing.assert_array_equal(v, outs[n][1])
        outs[n] = (p, v, d)
    np.testing.assert_array_equal(outs[9][0], outs[16][0][:9])  # the same team size: the same bits whatever else runs
    for n in (1, 2, 3, 5, 8):  # other team sizes: the row groups only regroup independent row fragments
        np.testing.assert_array_equal(outs[n][0], outs[16][0][:n])
    monkeypatch.setenv("NSG_TEAM_TRUNK", "0")
    old, _ = make(nsg, 4, 192, 16, precision="f16x3", seed=520)
    po, vo, do = old.compute_blocking(bb)
    assert old.last_plan()["waves_per_group"] != 8 and old.team_stats()["enabled"] == 0
    assert float(np.abs(po - outs[16][0]).max()) < 1e-4 and float(np.abs(vo - outs[16][1]).max()) < 1e-4


def test_team_sizes_and_member_override(nsg, monkeypatch):
    """Workgroups per board follow what fits the device (every member must be resident at once); NSG_TEAM_MEMBERS
    caps the row groups, is read per evaluator, and a value outside {16, 32, 48, 96} is refused."""
    ev, _ = make(nsg, 2, 256, 16, precision="f16m6", seed=530)
    cus = ev.info()["compute_units"]
    bb = nsg.synth.random_batch(16, 86, seed=531)
    ref = {}
    for n in (1, 2, 3, 5, 6, 8, 9, 16):
        ref[n] = ev.compute_blocking(bb[:n])
        want = next(16 * rg for rg in (6, 3, 2, 1) if n * 16 * rg <= cus)
        assert ev.team_stats()["members_last"] == want, (n, ev.team_stats())
    monkeypatch.setenv("NSG_TEAM_MEMBERS", "32")
    capped, _ = make(nsg, 2, 256, 16, precision="f16m6", seed=530)
    for n in (1, 5, 9):
        p, v, d = capped.compute_blocking(bb[:n])
        assert capped.team_stats()["members_last"] == (32 if n * 32 <= cus else 16)
        np.testing.assert_array_equal(p, ref[n][0])
    monkeypatch.setenv("NSG_TEAM_MAX_BATCH", "4")
    small, _ = make(nsg, 2, 256, 16, precision="f16m6", seed=530)
    small.compute_blocking(bb[:4])
    assert small.last_plan()["k_split"] == 8
    small.compute_blocking(bb[:5])
    assert small.last_plan()["k_split"] != 8


@pytest.mark.parametrize("path", ["blocking", "resident", "gather"])
def test_team_trunk_gives_up_and_the_batch_is_rerun(nsg, oracle, monkeypatch, path):
    """A team launch whose members wait for each other in vain (here: the launch is made ONE WORKGROUP SHORT through the
    test hook NSG_TEAM_FAULT_LAUNCHES; in the field: a device partition smaller than the grid, another process's
    persistent kernel) gives up after its bounded spins.  The call that waits for it re-runs the SAME batch on the
    per-layer kernels and succeeds -- nsg_await, and the device-resident read-backs too -- the evaluator keeps to those
    kernels afterwards and counts the event."""
    monkeypatch.setenv("NSG_TEAM_FAULT_LAUNCHES", "1")
    ev, blob = make(nsg, 2, 256, 8, precision="f16m6", seed=540)
    bb = nsg.synth.random_batch(8, 86, seed=541, garbage=True)
    ref = oracle.net(blob).evaluate(bb)
    assert ev.team_stats() == {"enabled": 1, "members_last": 0, "fallbacks": 0}
    if path == "blocking":
        out = ev.compute_blocking(bb[:5])
        check(out, tuple(r[:5] for r in ref), 2e-4)
    elif path == "resident":
        ev.upload_features(bb[:5])
        ev.forward_resident(5)
        out = ev.download_outputs(5)
        check(out, tuple(r[:5] for r in ref), 2e-4)
    else:
        rng = np.random.default_rng(3)
        off = (np.arange(6) * 40).astype(np.uint32)
        idx = rng.integers(0, 2187, size=200).astype(np.uint16)
        vals = np.empty(200, np.float32)
        win, draw = np.empty(5, np.float32), np.empty(5, np.float32)
        ev.compute_gather_blocking(bb[:5], idx, off, softmax=False, values=vals, win=win, draw=draw)
        want = np.concatenate([ref[0][b][idx[40 * b:40 * b + 40]] for b in range(5)])
        assert float(np.abs(vals - want).max()) < 2e-4 and float(np.abs(win - ref[1][:5]).max()) < 2e-4
    ts = ev.team_stats()
    assert ts["fallbacks"] == 1 and ts["enabled"] == 0 and ts["members_last"] == 48, ts
    assert ev.last_plan()["k_split"] != 8  # the re-run took the per-layer kernels
    # ... and so does every later batch; the statistics count the batch once
    p2, v2, d2 = ev.compute_blocking(bb)
    check((p2, v2, d2), ref, 2e-4)
    assert ev.last_plan()["k_split"] != 8 and ev.team_stats()["fallbacks"] == 1
    assert ev.stats()["batches"] == 2


def test_team_trunk_two_processes_one_device(nsg, tmp_path):
    """Two PROCESSES on one GPU, each evaluating batches of at most sixteen boards: their team launches would hold
    CUs the other's unscheduled members need.  One process owns the device's team token (an advisory file lock),
    the other keeps to the per-layer kernels; both finish, neither pays a give-up, both agree with each other."""
    import subprocess, sys, textwrap
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import importlib, json, os, sys, time
        import numpy as np
        sys.path.insert(0, sys.argv[1])
        nsg = importlib.import_module("nshogi-engine_amd")
        ev = nsg.Evaluator(0, 16, 86, precision="f16m6")
        ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(3, 256, seed=550, bn="random")))
        bb = nsg.synth.random_batch(16, 86, seed=551)
        open(sys.argv[2] + ".ready", "w").close()
        while not os.path.exists(sys.argv[3] + ".ready"):
            time.sleep(0.01)
        t0 = time.time()
        acc = 0.0
        for i in range(150):
            n = 1 + (i * 7) % 16
            p, v, d = ev.compute_blocking(bb[:n])
            acc += float(p[0, :8].sum())
        print(json.dumps({"team": ev.team_stats(), "first": ev.compute_blocking(bb[:3])[0][:, :4].tolist(),
                          "seconds": time.time() - t0}))
    """))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tags = [str(tmp_path / "a"), str(tmp_path / "b")]
    # (this pytest process may itself hold the device's lock through evaluators not collected yet: the two workers
    # coordinate through a lock directory of their own)
    env = dict(os.environ, NSG_TEAM_LOCK_DIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, str(script), root, tags[i], tags[1 - i]], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for i in range(2)]
    outs = []
    for pr in procs:
        so, se = pr.communicate(timeout=600)
        assert pr.returncode == 0, se[-2000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    enabled = sorted(o["team"]["enabled"] for o in outs)
    assert enabled == [-1, 1], outs
    assert all(o["team"]["fallbacks"] == 0 for o in outs), outs
    # team trunk (f16x3 arithmetic) in one process, per-layer f16x3 small tiles in the other: the summation order differs
    assert float(np.abs(np.array(outs[0]["first"]) - np.array(outs[1]["first"])).max()) < 1e-4


@pytest.mark.parametrize("batch", [17, 30, 44, 64, 85])
def test_f16m6_three_way_k_split_192_channels(nsg, oracle, monkeypatch, batch):
    """192 channels are three chunk pairs: up to CUs/3 boards run one 64-channel group per workgroup whose three waves
    take one pair each (all chunk tiles of the board resident in LDS), the rows over as many workgroups as fit one
    round; the stem (two pairs) runs two-wave workgroups.  The MX arithmetic instead of the f16x3 small tiles these
    batches ran before (NSG_KSPLIT3=0).  Against the oracle and the f16x3 evaluator."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    if batch * 3 > cus:
        pytest.skip("batch range of this plan depends on the CU count")
    ev, blob = make(nsg, 3, 192, batch, precision="f16m6", seed=560)
    bb = nsg.synth.random_batch(batch, 86, seed=561, garbage=True)
    p, v, d = ev.compute_blocking(bb)
    plan = ev.last_plan()
    assert plan["trunk_precision"] == "f16m6" and plan["k_split"] == 3 and plan["waves_per_group"] == 3, plan
    assert plan["row_split"] == (6 if batch * 18 <= cus else 3 if batch * 9 <= cus else 2 if batch * 6 <= cus else 1)
    idx = sorted({0, batch // 2, batch - 1})
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), TOL)
    x3, _ = make(nsg, 3, 192, batch, precision="f16x3", seed=560)
    p3, v3, d3 = x3.compute_blocking(bb)
    assert float(np.abs(p - p3).max()) < TOL and float(np.abs(v - v3).max()) < TOL
    p2, v2, d2 = ev.compute_blocking(bb[::-1].copy())  # deterministic, slot-independent
    np.testing.assert_array_equal(p2[::-1], p)
    monkeypatch.setenv("NSG_KSPLIT3", "0")
    old, _ = make(nsg, 3, 192, batch, precision="f16m6", seed=560)
    po, vo, do = old.compute_blocking(bb)
    assert old.last_plan()["k_split"] != 3
    assert float(np.abs(p - po).max()) < TOL


@pytest.mark.parametrize("channels,batch", [(256, 65), (256, 101), (256, 128), (256, 17), (256, 24), (256, 40), (256, 64),
                                            (192, 17), (192, 30), (192, 64), (192, 85)])
def test_cooperative_trunk_mid_batches(nsg, oracle, monkeypatch, channels, batch):
    """The K-split plans of an f16m6 evaluator -- several workgroups per board: two 128-channel halves at 65 ... CUs/2
    boards, four 64-channel groups (x 1 / 2 / 3 / 6 row groups) at 17 ... CUs/4, three (192 channels) at 17 ... CUs/3 --
    as ONE launch for all 3x3 layers: a workgroup waits for the other workgroups of ITS board only, not for the slowest of
    the whole grid 41 times per forward (mfma_tile.h, coopTrunkKernel; the stem, which has fewer chunk pairs than the
    four- and three-way splits need, keeps its own launch).  Same tile code as the per-layer kernels: bit-identical to
    them (NSG_COOP_TRUNK=0), against the oracle, deterministic and slot-independent."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    if channels == 256:
        ks = 4 if batch * 4 <= cus else 2
        if ks == 2 and not (batch * 2 <= cus):
            pytest.skip("batch range of this plan depends on the CU count")
    else:
        ks = 3
        if batch * 3 > cus:
            pytest.skip("batch range of this plan depends on the CU count")
    # by default the cooperative launch is taken where it measured faster (256 channels: at most eight members per board)
    auto, _ = make(nsg, 3, channels, batch, precision="f16m6", seed=610)
    auto.compute_blocking(nsg.synth.random_batch(batch, 86, seed=611, garbage=True))
    members = {2: 2, 3: 3, 4: 4}[ks] * auto.last_plan()["row_split"]
    # the members of a board share an XCD (blockIdx.x picks it): ceil(batch / 8) boards' members must fit an XCD's CUs
    fits = (batch + 7) // 8 * members <= cus // 8
    assert auto.last_launch_kind()[0] == ("coop" if (channels == 192 or members <= 8) and fits else "per_layer"), (auto.last_launch_kind(), members)
    monkeypatch.setenv("NSG_COOP_TRUNK", "1")  # ... here: every plan that has a cooperative form
    ev, blob = make(nsg, 3, channels, batch, precision="f16m6", seed=610)
    bb = nsg.synth.random_batch(batch, 86, seed=611, garbage=True)
    p, v, d = ev.compute_blocking(bb)
    if not fits:
        assert ev.last_launch_kind()[0] == "per_layer"
        return
    assert ev.last_launch_kind() == ("coop", 1) and ev.last_plan()["k_split"] == ks, (ev.last_launch_kind(), ev.last_plan())
    assert ev.team_stats()["fallbacks"] == 0
    idx = sorted({0, batch // 2, batch - 1})
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), TOL)
    for _ in range(3):  # launch after launch, and with the boards in other slots
        p2, v2, d2 = ev.compute_blocking(bb[::-1].copy())
        np.testing.assert_array_equal(p2[::-1], p)
        np.testing.assert_array_equal(v2[::-1], v)
    ev.upload_features(bb)  # device-resident, queued behind each other
    for _ in range(4):
        ev.forward_resident(batch)
    pr, vr, dr = ev.download_outputs(batch)
    np.testing.assert_array_equal(pr, p)
    monkeypatch.setenv("NSG_COOP_TRUNK", "0")
    per, _ = make(nsg, 3, channels, batch, precision="f16m6", seed=610)
    pp, vp, dp = per.compute_blocking(bb)
    # (the same plan, but for 17-21 boards of a 256-channel net: three row groups per layer, two as one launch -- row groups
    # do not change a bit of the result)
    same = {k: v for k, v in per.last_plan().items() if k != "row_split"} == {k: v for k, v in ev.last_plan().items() if k != "row_split"}
    assert per.last_launch_kind() == ("per_layer", 0) and same, (per.last_plan(), ev.last_plan())
    np.testing.assert_array_equal(pp, p)
    np.testing.assert_array_equal(vp, v)
    np.testing.assert_array_equal(dp, d)


def test_cooperative_trunk_gives_up_and_the_batch_is_rerun(nsg, oracle, monkeypatch):
    """The cooperative trunk's safety net (the team trunk's: test_team_trunk_gives_up_and_the_batch_is_rerun): a
    member that never publishes (test hook) makes its board's other member give up; the waiting call re-runs the batch
    on the per-layer kernels and succeeds, later batches keep to them."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    batch = max(2, cus // 2 - 5)
    monkeypatch.setenv("NSG_TEAM_FAULT_LAUNCHES", "1")
    ev, blob = make(nsg, 2, 256, batch, precision="f16m6", seed=620)
    bb = nsg.synth.random_batch(batch, 86, seed=621)
    p, v, d = ev.compute_blocking(bb)
    assert ev.team_stats()["fallbacks"] == 1 and ev.last_launch_kind() == ("per_layer", 0)
    idx = [0, batch - 1]
    check((p[idx], v[idx], d[idx]), oracle.net(blob).evaluate(bb[idx]), TOL)
    p2, v2, d2 = ev.compute_blocking(bb)
    np.testing.assert_array_equal(p2, p)
    assert ev.team_stats()["fallbacks"] == 1 and ev.stats()["batches"] == 2


def test_cooperative_trunk_flag_values_run_out_and_start_again(nsg, monkeypatch):
    """The cooperative trunk's flags are not cleared between launches: their values count on from launch to launch (24
    bits under the XCC_ID byte) and the array is cleared when they run out.  Launches on either side of that point, and a
    launch with another member count in between, give the per-layer kernels' outputs."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    big, small = max(2, cus // 2 - 9), max(2, cus // 8)
    monkeypatch.setenv("NSG_COOP_TRUNK", "0")
    per, blob = make(nsg, 2, 256, big, precision="f16m6", seed=640)
    bb = nsg.synth.random_batch(big, 86, seed=641)
    want_big, want_small = per.compute_blocking(bb), per.compute_blocking(bb[:small])
    monkeypatch.delenv("NSG_COOP_TRUNK")
    monkeypatch.setenv("NSG_COOP_FLAG_BASE", str((1 << 24) - 600))  # the fourth launch clears the array and starts at zero
    ev, _ = make(nsg, 2, 256, big, precision="f16m6", seed=640)
    for i in range(8):
        n, want = (small, want_small) if i % 3 == 1 else (big, want_big)
        got = ev.compute_blocking(bb[:n])
        assert ev.last_launch_kind()[0] == "coop"
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
    assert ev.team_stats()["fallbacks"] == 0


def test_cooperative_trunk_members_on_different_xcds_are_noticed(nsg, oracle, monkeypatch):
    """The cooperative trunk hands a board's rows from member to member through the L2 of ONE XCD (plain stores,
    L1-bypassing loads): every member publishes its XCC_ID with its flag and every poll compares it with the poller's own.
    A member that reports another XCD (test hook) -- a placement the hand-off cannot use -- makes its board give up; the
    waiting call re-runs the batch on the per-layer kernels, bit-identical, and later batches keep to them."""
    probe = nsg.Evaluator(0, 1, 86)
    cus = probe.info()["compute_units"]
    del probe
    batch = max(2, cus // 4 - 3)
    ok, blob = make(nsg, 2, 256, batch, precision="f16m6", seed=630)
    bb = nsg.synth.random_batch(batch, 86, seed=631)
    p0, v0, d0 = ok.compute_blocking(bb)
    assert ok.last_launch_kind()[0] == "coop" and ok.team_stats()["fallbacks"] == 0  # this GPU deals the members as assumed
    monkeypatch.setenv("NSG_COOP_FAULT_XCC_LAUNCHES", "1")
    ev, _ = make(nsg, 2, 256, batch, precision="f16m6", seed=630)
    p, v, d = ev.compute_blocking(bb)
    assert ev.team_stats()["fallbacks"] == 1 and ev.last_launch_kind() == ("per_layer", 0)
    np.testing.assert_array_equal(p, p0)
    np.testing.assert_array_equal(v, v0)
    p2, v2, d2 = ev.compute_blocking(bb)
    np.testing.assert_array_equal(p2, p0)
    assert ev.team_stats()["fallbacks"] == 1 and ev.stats()["batches"] == 2
