"""Synthetic ml::FeatureBitboard batches (SURVEY.md 8a a1, 8d "Synthetic inputs").

A feature bitboard is 16 bytes = (lo, hi) uint64:
  lo bits 0..62  = squares 0..62          hi bits 0..17  = squares 63..80
  hi bit 24      = rotate-180 flag         hi bits 32..63 = f32 bit pattern of the plane value
(layout read off /root/reference/src/cuda/extractbit.cu:20-37).  Real planes need
libnshogi's FeatureStack, which is absent, so the bench and the tests use seeded
synthetic planes of the same shape and statistics: 28 sparse piece planes,
52 all-or-nothing hand planes, 6 global planes of which 4 carry a scalar value.
"""
import numpy as np

SEED = 20240203  # the reference's own test seed (src/test/test_extractbit.cc:72)
ONE = np.uint64(0x3F800000)


def pack(square_bits, rotate, value_f32):
    """square_bits: bool [..., 81]; rotate: bool [...]; value_f32: float32 [...] -> uint64 [..., 2]"""
    sb = np.asarray(square_bits, dtype=np.uint64)
    lo = np.zeros(sb.shape[:-1], dtype=np.uint64)
    hi = np.zeros(sb.shape[:-1], dtype=np.uint64)
    for s in range(63):
        lo |= sb[..., s] << np.uint64(s)
    for s in range(63, 81):
        hi |= sb[..., s] << np.uint64(s - 63)
    hi |= np.asarray(rotate, dtype=np.uint64) << np.uint64(24)
    vbits = np.asarray(value_f32, dtype=np.float32).view(np.uint32).astype(np.uint64)
    hi |= vbits << np.uint64(32)
    return np.stack([lo, hi], axis=-1)


def random_batch(batch, channels=86, seed=SEED, distinct=True, garbage=False):
    """uint64 [batch, channels, 2].  distinct=False replicates one position
    (the reference's own benchmark does that, src/bench/batchsize.cc:47-59)."""
    rng = np.random.default_rng(seed)
    nb = batch if distinct else 1
    bits = np.zeros((nb, channels, 81), dtype=bool)
    value = np.ones((nb, channels), dtype=np.float32)
    n_piece = min(28, channels)
    dens = rng.integers(1, 21, size=(nb, n_piece))
    bits[:, :n_piece] = rng.random((nb, n_piece, 81)) < (dens[..., None] / 81.0)
    if channels > n_piece:
        flags = rng.random((nb, channels - n_piece)) < 0.3
        bits[:, n_piece:] = flags[..., None]
    if channels >= 4:  # scalar planes: all squares set, value uniform [0,1)
        bits[:, channels - 4:] = True
        value[:, channels - 4:] = rng.random((nb, 4), dtype=np.float32)
    rotate = (np.arange(nb) % 2 == 1)[:, None] & np.ones((1, channels), dtype=bool)
    bb = pack(bits, rotate, value)
    if garbage:  # bits the kernel must ignore: lo 63, hi 18..23 and 25..31
        junk = rng.integers(0, 2 ** 63, size=(nb, channels), dtype=np.uint64)
        bb[..., 0] |= (junk & np.uint64(1)) << np.uint64(63)
        bb[..., 1] |= junk & np.uint64(0xFEFC0000)
    if not distinct:
        bb = np.repeat(bb, batch, axis=0)
    return np.ascontiguousarray(bb)


def expand_reference(bb, channels_first=True):
    """Independent numpy restatement of the plane expansion (used to build the
    golden fixtures and to cross-check the C oracle)."""
    bb = np.asarray(bb, dtype=np.uint64)
    lo, hi = bb[..., 0], bb[..., 1]
    rotate = ((hi >> np.uint64(24)) & np.uint64(1)).astype(bool)
    value = (hi >> np.uint64(32)).astype(np.uint32)
    out = np.zeros(bb.shape[:-1] + (81,), dtype=np.uint32)
    for bit in range(81):
        target = np.where(rotate, 80 - bit, bit)
        use_hi = target >= 63
        shift = np.where(use_hi, target - 63, target).astype(np.uint64)
        word = np.where(use_hi, hi, lo)
        on = ((word >> shift) & np.uint64(1)).astype(bool)
        out[..., bit] = np.where(on, value, np.uint32(0))
    planes = out.view(np.float32)  # [B, C, 81]
    if not channels_first:
        planes = np.ascontiguousarray(np.swapaxes(planes, -1, -2))  # [B, 81, C]
    return planes
