"""Parity at the BENCHMARK's arithmetic on the positions that are actually evaluated: the initial
position (the reference benchmark's input in every slot, /root/reference/src/bench/batchsize.cc:47-59)
and positions of real games (what self-play feeds the evaluator) -- saturated hand / scalar planes,
sparse boards -- not the seeded random bitboards of the other parity tests.  A sample of >= 32 boards,
stratified over the plies of the games, against the CPU oracle at the north_star's 1e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3  # north_star: "within 1e-3 fp32"


def _batch(nsg, batch):
    """Half the slots hold the initial position, the other half distinct positions of random-playout
    games in game order (ply 0, 1, 2, ... of game after game)."""
    half = batch // 2
    return np.ascontiguousarray(np.concatenate([nsg.positions.startpos_batch(half),
                                                nsg.positions.game_positions(batch - half, seed=20240203)]))


@pytest.mark.parametrize("blocks,channels,batch,precision,sample", [
    (20, 256, 512, "f16m6", 40),   # bench.py's line: BASELINE configs[2]
    (20, 256, 512, "f16m8", 32),
    (20, 256, 128, "f16m6", 32),   # the batch self-play runs at (K-split tiles)
    (10, 192, 64, "f16m6", 32),    # BASELINE configs[1]
])
def test_real_positions_at_the_benchmark_arithmetic(nsg, oracle, blocks, channels, batch, precision, sample):
    blob = nsg.weights.to_blob(nsg.weights.make_random(blocks, channels, seed=0, bn="identity"))  # bench.py's weights
    ev = nsg.Evaluator(0, batch, 86, precision=precision)
    ev.load_memory(blob)
    bb = _batch(nsg, batch)
    p, v, d = ev.compute_blocking(bb)
    half = batch // 2
    # every slot holding the initial position returns the same bits (batchsize.cc:52-59 replicates it)
    assert (p[:half] == p[0]).all() and (v[:half] == v[0]).all() and (d[:half] == d[0]).all()
    # the initial position + a stratified sample of the game positions (evenly spaced over the plies)
    idx = np.unique(np.concatenate([[0], np.linspace(half, batch - 1, sample - 1).round().astype(int)]))
    assert len(idx) >= 32
    po, vo, do = oracle.net(blob).evaluate_parallel(bb[idx])
    err = (float(np.abs(p[idx] - po).max()), float(np.abs(v[idx] - vo).max()), float(np.abs(d[idx] - do).max()))
    print(f"{blocks}x{channels} B={batch} {precision}: max|err| policy {err[0]:.2e} value {err[1]:.2e} draw {err[2]:.2e} "
          f"on {len(idx)} boards (logit range {np.abs(po).max():.2f})")
    assert max(err) <= TOL, err
    assert np.isfinite(p).all() and ((v >= 0) & (v <= 1)).all() and ((d >= 0) & (d <= 1)).all()
    ev.close()
