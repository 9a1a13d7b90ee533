"""HIP feature-plane expansion (K1/K2 replacement) vs the oracle and the golden
fixtures, through the C ABI (nsg_extract_bits).  Bit-exact: the kernel is an
integer select of a bit pattern (src/cuda/extractbit.cu:19-37)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_extract(nsg, bb, channels_first):
    b, c = bb.shape[0], bb.shape[1]
    src = torch.from_numpy(bb.view(np.int64).copy()).cuda()
    dst = torch.full((b * c * 81,), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    nsg.extract_bits(dst.data_ptr(), src.data_ptr(), b, c, channels_first,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    shape = (b, c, 81) if channels_first else (b, 81, c)
    return dst.cpu().numpy().reshape(shape)


@pytest.mark.parametrize("cf", [True, False])
def test_golden_g2(nsg, golden_dir, cf):
    g = np.load(f"{golden_dir}/extract_g2.npz")
    out = run_extract(nsg, g["bitboards"], cf)
    np.testing.assert_array_equal(out.view(np.uint32), g["nchw_bits" if cf else "nhwc_bits"])


@pytest.mark.parametrize("batch,channels", [(1, 86), (3, 86), (64, 86), (513, 86), (5, 93), (2, 1),
                                            (1, 1024), (7, 63), (1, 65)])
@pytest.mark.parametrize("cf", [True, False])
def test_vs_oracle(nsg, oracle, batch, channels, cf):
    bb = nsg.synth.random_batch(batch, channels, seed=batch * 31 + channels, garbage=True)
    out = run_extract(nsg, bb, cf)
    ref = oracle.extract_bits(bb, cf)
    np.testing.assert_array_equal(out.view(np.uint32), ref.view(np.uint32))


def test_reference_test_shape(nsg, oracle):
    """The reference's own test runs batch 1, C = FeatureType::size() at every
    ply of a game (src/test/test_extractbit.cc:26-63): many batch-1 launches."""
    for ply in range(40):
        bb = nsg.synth.random_batch(1, 86, seed=20240203 + ply, garbage=(ply % 2 == 0))
        for cf in (True, False):
            np.testing.assert_array_equal(run_extract(nsg, bb, cf).view(np.uint32),
                                          oracle.extract_bits(bb, cf).view(np.uint32))


def test_walk_a_random_game_like_the_reference_test(nsg, golden_dir):
    """src/test/test_extractbit.cc:26-91 restated on real positions: one random game from the
    initial position (mt19937_64(20240203), the committed fixture), and at EVERY ply
    nsg_extract_bits at batch 1 in both layouts; all C*81 values compared with planes written
    from the board (tests/shogi_ref.py reads the SFEN), not from the bitboards."""
    import os
    import shogi_ref
    g = np.load(os.path.join(golden_dir, "game_20240203.npz"))
    assert len(g["sfens"]) > 200
    for sfen, bb in zip(g["sfens"], g["bitboards"]):
        want = shogi_ref.expected_planes(str(sfen), 1024, 0.5)  # MaxPly 1024 (test_extractbit.cc:31-35)
        got_cf = run_extract(nsg, bb[None], True)[0]
        got_cl = run_extract(nsg, bb[None], False)[0]
        np.testing.assert_array_equal(got_cf, want, err_msg=str(sfen))
        np.testing.assert_array_equal(got_cl, want.T, err_msg=str(sfen))


def test_walk_fresh_random_games_whole_batch(nsg):
    """Further games played now by the build's rules core (perft features), expanded as one
    batch per game in both layouts."""
    import test_features
    import shogi_ref
    for seed in (1, 2):
        rows = test_features.dump(1, seed, 1024, 0.5, 1024)
        bb = np.stack([r[1] for r in rows])
        want = np.stack([shogi_ref.expected_planes(r[0]) for r in rows])
        np.testing.assert_array_equal(run_extract(nsg, bb, True), want)
        np.testing.assert_array_equal(run_extract(nsg, bb, False), np.swapaxes(want, 1, 2))


def test_full_size_properties(nsg):
    """B = 1024 (config 5): rotate-involution and popcount properties that do
    not need the oracle at full size."""
    bb = nsg.synth.random_batch(1024, 86, seed=1)
    out = run_extract(nsg, bb, True)
    # flipping the rotate flag reverses every plane
    flipped = bb.copy()
    flipped[..., 1] ^= np.uint64(1 << 24)
    out2 = run_extract(nsg, flipped, True)
    np.testing.assert_array_equal(out2, out[..., ::-1])
    # number of non-zero outputs == popcount of the 81 square bits (value != 0 planes)
    lo, hi = bb[..., 0], bb[..., 1]
    pop = np.zeros(lo.shape, dtype=np.int64)
    for s in range(63):
        pop += ((lo >> np.uint64(s)) & np.uint64(1)).astype(np.int64)
    for s in range(18):
        pop += ((hi >> np.uint64(s)) & np.uint64(1)).astype(np.int64)
    nz = (out.view(np.uint32) != 0).sum(axis=-1)
    value_nonzero = (hi >> np.uint64(32)) != 0
    np.testing.assert_array_equal(nz, np.where(value_nonzero, pop, 0))
    # NHWC is the transpose
    np.testing.assert_array_equal(run_extract(nsg, bb, False), np.swapaxes(out, 1, 2))


def test_bad_arguments(nsg):
    with pytest.raises(nsg.NsgError):
        nsg.extract_bits(0, 0, 1, 86)
    t = torch.zeros(16, device="cuda")
    with pytest.raises(nsg.NsgError):
        nsg.extract_bits(t.data_ptr(), t.data_ptr(), 0, 86)
    with pytest.raises(nsg.NsgError):  # extractbit.cu:91: NHWC needs C <= 1024
        nsg.extract_bits(t.data_ptr(), t.data_ptr(), 1, 1025, channels_first=False)
