"""Oracle (C restatement of src/cuda/extractbit.cu) vs the golden fixtures and
vs the independent numpy restatement.  CPU only."""
import numpy as np
import pytest


def _golden(golden_dir):
    return np.load(f"{golden_dir}/extract_g2.npz")


def test_oracle_matches_golden_nchw(oracle, golden_dir):
    g = _golden(golden_dir)
    out = oracle.extract_bits(g["bitboards"], channels_first=True)
    assert out.shape == (g["bitboards"].shape[0], 86, 81)
    np.testing.assert_array_equal(out.view(np.uint32), g["nchw_bits"])


def test_oracle_matches_golden_nhwc(oracle, golden_dir):
    g = _golden(golden_dir)
    out = oracle.extract_bits(g["bitboards"], channels_first=False)
    np.testing.assert_array_equal(out.view(np.uint32), g["nhwc_bits"])


def test_branch_cases_by_hand(oracle, nsg):
    """Every branch of extractbit.cu:19-37 checked against hand-written expectations."""
    synth = nsg.synth
    def one(squares, rotate, value):
        bits = np.zeros(81, dtype=bool)
        bits[list(squares)] = True
        bb = synth.pack(bits, rotate, np.float32(value)).reshape(1, 1, 2)
        return oracle.extract_bits(bb)[0, 0]
    # square 0 un-rotated -> index 0; rotated -> index 80
    assert one([0], False, 1.0).tolist() == [1.0] + [0.0] * 80
    assert one([0], True, 1.0).tolist() == [0.0] * 80 + [1.0]
    # lo/hi boundary: squares 62 (lo bit 62) and 63 (hi bit 0)
    e = one([62, 63], False, 0.5)
    assert e[62] == 0.5 and e[63] == 0.5 and e.sum() == 1.0
    e = one([62, 63], True, 0.5)
    assert e[80 - 62] == 0.5 and e[80 - 63] == 0.5 and e.sum() == 1.0
    # last square
    assert one([80], False, 2.0)[80] == 2.0 and one([80], True, 2.0)[0] == 2.0
    # value 0 -> all zeros even with bits set
    assert not one(range(81), False, 0.0).any()


def test_ignored_bits(oracle, nsg):
    """Only hi bits 0..17, 24 and 32..63 and lo bits 0..62 are read."""
    clean = nsg.synth.random_batch(3, 86, seed=5, garbage=False)
    dirty = clean.copy()
    dirty[..., 0] |= np.uint64(1) << np.uint64(63)
    dirty[..., 1] |= np.uint64(0xFEFC0000)
    a = oracle.extract_bits(clean).view(np.uint32)
    b = oracle.extract_bits(dirty).view(np.uint32)
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("batch,channels", [(1, 86), (7, 86), (3, 93), (2, 1), (1, 1024)])
def test_oracle_vs_numpy_random(oracle, nsg, batch, channels):
    bb = nsg.synth.random_batch(batch, channels, seed=batch * 1000 + channels, garbage=True)
    for cf in (True, False):
        a = oracle.extract_bits(bb, channels_first=cf).view(np.uint32)
        b = nsg.synth.expand_reference(bb, channels_first=cf).view(np.uint32)
        np.testing.assert_array_equal(a, b)


def test_nhwc_is_transpose_of_nchw(oracle, nsg):
    bb = nsg.synth.random_batch(4, 86, seed=9)
    a = oracle.extract_bits(bb, True)
    b = oracle.extract_bits(bb, False)
    np.testing.assert_array_equal(np.swapaxes(a, 1, 2), b)
