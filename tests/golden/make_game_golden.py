"""Generates tests/golden/game_20240203.npz (run from the repo root:
`python tests/golden/make_game_golden.py`).

The reference's only hot-path test (src/test/test_extractbit.cc:26-91) plays one random
game from the initial position with std::mt19937_64(20240203), Moves[Mt() % Moves.size()],
up to 1024 plies, and at every ply compares the plane expansion of the position's feature
bitboards with FeatureStack::extract.  libnshogi (move generation order, feature stack) is
absent, so the game below is the same recipe on this build's rules core
(`perft features 1 20240203 1024 0.5 1024`): for every ply the SFEN, the 86 feature
bitboards the C++ feature builder wrote, and the legal moves with their policy indices.
The EXPECTED planes are not stored: tests rebuild them from the SFEN text
(tests/shogi_ref.py), so the fixture pins inputs and the C++ builder's bytes only.
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shogi_ref  # noqa: E402

PERFT = os.path.join(ROOT, "nshogi-engine_amd", "csrc", "selfplay", "perft")


def main():
    out = subprocess.run([PERFT, "features", "1", "20240203", "1024", "0.5", "1024"],
                         capture_output=True, text=True, check=True).stdout
    rows = [shogi_ref.parse_dump_line(ln) for ln in out.splitlines()]
    sfens = np.array([r[0] for r in rows])
    bb = np.stack([r[1] for r in rows])
    moves = np.array([" ".join(f"{u}:{i}" for u, i in r[2]) for r in rows])
    path = os.path.join(ROOT, "tests", "golden", "game_20240203.npz")
    np.savez_compressed(path, sfens=sfens, bitboards=bb, moves=moves)
    print(path, len(rows), "plies", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
