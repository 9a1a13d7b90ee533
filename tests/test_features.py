"""csrc/shogi/features.cc -- the 86 feature planes (FeatureType::constructAt,
/root/reference/src/evaluate/preset.h:20-66, src/selfplay/evaluationworker.cc:87-92) and the
policy move index (ml::getMoveIndex, src/selfplay/frame.cc:102-105,
src/mcts/feedworker.cc:120-125) -- against tests/shogi_ref.py, which rebuilds both from the
SFEN text alone, over random-playout games from the initial position (the shape of the
reference's only hot-path test, src/test/test_extractbit.cc:66-91)."""
import os
import subprocess

import numpy as np
import pytest

import shogi_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PERFT = os.path.join(ROOT, "nshogi-engine_amd", "csrc", "selfplay", "perft")


def perft(*args):
    if not os.path.exists(PERFT):
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([PERFT] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def dump(games, seed, max_ply=1024, black_draw=0.5, stop=400):
    lines = perft("features", games, seed, max_ply, black_draw, stop).splitlines()
    return [shogi_ref.parse_dump_line(ln) for ln in lines]


@pytest.fixture(scope="module")
def games(nsg):
    return dump(6, 20240203)  # the reference test's own seed (test_extractbit.cc:72)


def test_planes_equal_the_board_restatement(nsg, oracle, games):
    """Every value of every plane at every ply: expansion of the C++ bitboards (oracle =
    extractbit.cu restated) == planes written from the SFEN."""
    assert len(games) > 600
    bb = np.stack([g[1] for g in games])
    got = oracle.extract_bits(bb, True)  # [P, 86, 81]
    np.testing.assert_array_equal(got, nsg.synth.expand_reference(bb, True))
    for (sfen, _, _), planes in zip(games, got):
        want = shogi_ref.expected_planes(sfen)
        bad = np.argwhere(planes != want)
        assert bad.size == 0, f"{sfen}: first mismatch plane/square {bad[0]}"
    # channels-last is the transpose
    np.testing.assert_array_equal(oracle.extract_bits(bb[:50], False), np.swapaxes(got[:50], 1, 2))


@pytest.mark.parametrize("max_ply,black_draw", [(512, 0.25), (320, 1.0), (65535, 0.0)])
def test_scalar_planes_follow_the_state_config(oracle, max_ply, black_draw):
    """Progress = ply/MaxPly, ProgressUnit = 1/MaxPly, My/OpDrawValue per side to move
    (StateConfig of selfplay/worker.cc:132-150)."""
    for sfen, bb, _ in dump(1, 7, max_ply, black_draw, 60):
        planes = oracle.extract_bits(bb[None], True)[0]
        want = shogi_ref.expected_planes(sfen, max_ply, black_draw)
        np.testing.assert_array_equal(planes, want, err_msg=sfen)
        side = shogi_ref.parse_sfen(sfen)[1]
        assert planes[84, 0] == np.float32(black_draw if side == 0 else 1.0 - black_draw)


def test_bitboard_fields_and_piece_counts(games):
    """Layout of ml::FeatureBitboard (extractbit.cu:20-37): nothing outside lo[0:63),
    hi[0:18), hi[24], hi[32:64); rotate flag <=> White to move, on every plane; piece planes
    hold exactly the pieces on the board; 40 pieces in all with the hands."""
    for sfen, bb, _ in games:
        board, side, hands, ply = shogi_ref.parse_sfen(sfen)
        lo, hi = bb[:, 0], bb[:, 1]
        assert not np.any(lo >> np.uint64(63)) and not np.any(hi & np.uint64(0xFEFC0000)), sfen
        assert np.all(((hi >> np.uint64(24)) & np.uint64(1)) == side), sfen
        pop = np.array([bin(int(a)).count("1") + bin(int(b) & 0x3FFFF).count("1") for a, b in bb])
        assert pop[:28].sum() == len(board)
        mine = sum(1 for c, _ in board.values() if c == side)
        assert pop[:14].sum() == mine and pop[14:28].sum() == len(board) - mine
        assert pop[5] == 1 and pop[19] == 1  # one king each
        assert len(board) + sum(sum(h.values()) for h in hands) == 40
        # hand planes are all-or-nothing thresholds: count = number of full planes per piece
        plane = 28
        for owner in (side, 1 - side):
            for letter, n in shogi_ref.HAND_ORDER:
                full = int((pop[plane:plane + n] == 81).sum())
                assert set(pop[plane:plane + n]) <= {0, 81} and full == min(hands[owner][letter], n), sfen
                plane += n
        # piece planes carry 1.0; the value word of the Progress plane is ply/1024
        assert np.all((hi[:82] >> np.uint64(32)) == 0x3F800000)
        assert np.uint32(hi[82] >> np.uint64(32)).view(np.float32) == np.float32(ply) / np.float32(1024)


def test_move_index_is_the_restated_map_and_injective(games):
    seen_classes = set()
    total = 0
    for sfen, _, moves in games:
        side = shogi_ref.parse_sfen(sfen)[1]
        idx = [i for _, i in moves]
        assert len(set(idx)) == len(idx), f"move index collides at {sfen}"
        assert all(0 <= i < shogi_ref.MOVE_INDEX_MAX for i in idx)
        for usi, i in moves:
            assert i == shogi_ref.move_index(side, usi), (sfen, usi)
            seen_classes.add(i // 81)
        total += len(moves)
    assert total > 20000 and seen_classes == set(range(27))  # every class occurs


def test_maximum_moves_position_indices():
    """The 593-move position: all indices distinct and below 27*81."""
    sfen = "R8/2K1S1SSk/4B4/9/9/9/9/9/1L1L1L3 b RBGSNLP3g3n17p 1"
    _, _, moves = shogi_ref.parse_dump_line(perft("featuresat", 1024, 0.5, sfen))
    assert len(moves) == 593 and len({i for _, i in moves}) == 593
    assert all(i == shogi_ref.move_index(0, u) for u, i in moves)


def test_colour_symmetry(oracle, games):
    """Exchanging colours and turning the board by 180 degrees changes nothing the network sees
    except the Black/White-to-move planes, and maps every move to the same policy index."""
    for sfen, bb, moves in games[3::17]:
        line = perft("featuresat", 1024, 0.5, shogi_ref.flip_sfen(sfen))
        fsfen, fbb, fmoves = shogi_ref.parse_dump_line(line)
        a = oracle.extract_bits(bb[None], True)[0]
        b = oracle.extract_bits(fbb[None], True)[0]
        np.testing.assert_array_equal(a[:80], b[:80], err_msg=sfen)
        np.testing.assert_array_equal(a[82:], b[82:], err_msg=sfen)
        np.testing.assert_array_equal(a[80], b[81])
        np.testing.assert_array_equal(a[81], b[80])
        assert {(shogi_ref.flip_usi(u), i) for u, i in moves} == set(fmoves), sfen


def test_golden_game_fixture(oracle, golden_dir):
    """tests/golden/game_20240203.npz (the reference test's recipe on this rules core): the
    feature builder still writes these bytes, and they expand to the planes of the SFENs."""
    g = np.load(os.path.join(golden_dir, "game_20240203.npz"))
    live = dump(1, 20240203, 1024, 0.5, 1024)
    assert [r[0] for r in live] == list(g["sfens"])
    np.testing.assert_array_equal(np.stack([r[1] for r in live]), g["bitboards"])
    assert [" ".join(f"{u}:{i}" for u, i in r[2]) for r in live] == list(g["moves"])
    got = oracle.extract_bits(g["bitboards"], True)
    for sfen, planes in zip(g["sfens"], got):
        np.testing.assert_array_equal(planes, shogi_ref.expected_planes(str(sfen)), err_msg=str(sfen))
    assert len(g["sfens"]) == 204 and str(g["sfens"][0]).startswith("lnsgkgsnl/1r5b1/ppppppppp/9/9/9/")
