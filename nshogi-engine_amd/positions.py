"""Real positions as feature bitboards, from the build's own rules core.

The reference's benchmark fills every batch slot with the feature stack of the initial
position (/root/reference/src/bench/batchsize.cc:47-59: `FeatureType(State, Config)` of
`getInitialState()` copied BatchSize times).  libnshogi is absent, so the planes come from
csrc/shogi/features.cc through the `perft` tool (built by __graft_entry__.build()):
`startpos_batch` is the benchmark's input, `game_positions` gives distinct positions of
random-playout games for the B-distinct variant.
"""
import os
import subprocess

import numpy as np

NUM_PLANES = 86
PERFT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "selfplay", "perft")
STARTPOS_SFEN = "lnsgkgsnl/1r5b1/ppppppppp/9/9/9/PPPPPPPPP/1B5R1/LNSGKGSNL b - 1"


def _run(*args):
    if not os.path.exists(PERFT):
        raise FileNotFoundError(f"{PERFT} is missing: build it with "
                                "`python -c 'import __graft_entry__ as g; g.build()'`")
    r = subprocess.run([PERFT] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError(f"perft {' '.join(map(str, args))} failed: {r.stderr[-500:]}")
    return r.stdout


def _bitboards(line):
    return np.frombuffer(bytes.fromhex(line.split("\t")[1]), dtype="<u8").reshape(NUM_PLANES, 2).copy()


def startpos_bitboards(max_ply=1024, black_draw=0.5):
    """uint64 [86, 2]: feature bitboards of the initial position."""
    return _bitboards(_run("featuresat", max_ply, black_draw, STARTPOS_SFEN))


def startpos_batch(batch, max_ply=1024, black_draw=0.5):
    """uint64 [batch, 86, 2]: the initial position in every slot (batchsize.cc:52-59)."""
    return np.ascontiguousarray(np.repeat(startpos_bitboards(max_ply, black_draw)[None], batch, axis=0))


def game_positions(count, seed=20240203, max_ply=1024, black_draw=0.5):
    """uint64 [count, 86, 2]: distinct positions visited by random-playout games from the
    initial position (Moves[Mt() % size], the recipe of src/test/test_extractbit.cc:66-91)."""
    out, seen, games = [], set(), 0
    while len(out) < count and games < 64:
        games += 4
        for line in _run("features", 4, seed + games, max_ply, black_draw, 400).splitlines():
            sfen = line.split("\t", 1)[0].rsplit(" ", 1)[0]
            if sfen in seen:
                continue
            seen.add(sfen)
            out.append(_bitboards(line))
            if len(out) == count:
                break
    if len(out) < count:
        raise RuntimeError(f"only {len(out)} distinct positions found")
    return np.ascontiguousarray(np.stack(out))
