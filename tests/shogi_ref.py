"""Independent restatement -- from the SFEN text alone -- of what the self-play path feeds
the evaluator: the 86 feature planes of evaluate::preset::SimpleFeatures
(/root/reference/src/evaluate/preset.h:20-66) as the network sees them after the
plane expansion (src/cuda/extractbit.cu:19-37), and the policy index of a USI move
(role of ml::getMoveIndex<ChannelsFirst>, src/selfplay/frame.cc:102-105,
src/mcts/feedworker.cc:120-125).

TEST INFRASTRUCTURE: used by tests/ only, to check csrc/shogi/features.cc (the product)
against an expectation that is written from the board, not from the bitboards
(the shape of src/test/test_extractbit.cc:26-63, where libnshogi's FeatureStack::extract
plays this role).  libnshogi is absent: plane and index SEMANTICS are this build's reading
of the feature names -- parity with libnshogi unpinned; what is pinned is
that the C++ feature builder, the HIP expansion and this restatement agree.

Conventions of the build's shogi core (csrc/shogi/shogi.h): square = file*9 + rank,
file 0 = "1", rank 0 = "a"; Black moves toward rank 0.
"""
import numpy as np

PIECES = "PLNSBRGK"  # sfen letters
# preset.h order of the 14 board planes: Pawn Lance Knight Silver Gold King Bishop Rook,
# then +P +L +N +S Horse Dragon
BOARD_PLANE = {"P": 0, "L": 1, "N": 2, "S": 3, "G": 4, "K": 5, "B": 6, "R": 7,
               "+P": 8, "+L": 9, "+N": 10, "+S": 11, "+B": 12, "+R": 13}
HAND_ORDER = [("P", 6), ("L", 4), ("N", 4), ("S", 4), ("G", 4), ("B", 2), ("R", 2)]
NUM_PLANES = 86
MOVE_INDEX_MAX = 2187


def parse_sfen(sfen):
    """-> (board: dict square -> (color, kind) with color 0 = Black, kind like "P" or "+R";
    side: 0/1; hands: [dict letter -> count] per colour; ply: moves played so far)."""
    parts = sfen.split()
    board = {}
    for rank, row in enumerate(parts[0].split("/")):
        col, promo = 0, ""
        for ch in row:
            if ch == "+":
                promo = "+"
            elif ch.isdigit():
                col += int(ch)
            else:
                color = 1 if ch.islower() else 0
                board[(8 - col) * 9 + rank] = (color, promo + ch.upper())
                promo = ""
                col += 1
        assert col == 9, sfen
    side = 1 if parts[1] == "w" else 0
    hands = [dict.fromkeys("PLNSGBR", 0), dict.fromkeys("PLNSGBR", 0)]
    if parts[2] != "-":
        count = ""
        for ch in parts[2]:
            if ch.isdigit():
                count += ch
            else:
                hands[1 if ch.islower() else 0][ch.upper()] += int(count) if count else 1
                count = ""
    ply = int(parts[3]) - 1 if len(parts) > 3 else 0
    return board, side, hands, ply


def expected_planes(sfen, max_ply=1024, black_draw=0.5):
    """float32 [86, 81]: the planes in the side to move's own orientation (what the plane
    expansion writes in NCHW once the rotate flag is applied)."""
    board, side, hands, ply = parse_sfen(sfen)
    out = np.zeros((NUM_PLANES, 81), dtype=np.float32)
    for sq, (color, kind) in board.items():
        seen_from_mover = 80 - sq if side == 1 else sq
        out[(0 if color == side else 14) + BOARD_PLANE[kind], seen_from_mover] = 1.0
    plane = 28
    for owner in (side, 1 - side):
        for letter, n in HAND_ORDER:
            for k in range(1, n + 1):
                if hands[owner][letter] >= k:
                    out[plane] = 1.0
                plane += 1
    assert plane == 80
    out[80 + side] = 1.0
    out[82] = np.float32(ply) / np.float32(max_ply)
    out[83] = np.float32(1.0) / np.float32(max_ply)
    white_draw = np.float32(1.0) - np.float32(black_draw)
    out[84] = np.float32(black_draw) if side == 0 else white_draw
    out[85] = white_draw if side == 0 else np.float32(black_draw)
    return out


def usi_square(text):
    return (ord(text[0]) - ord("1")) * 9 + (ord(text[1]) - ord("a"))


# (d_file, d_rank) of one step, as seen by the mover, for classes 0..9
_DIRS = {(0, -1): 0, (-1, -1): 1, (1, -1): 2, (-1, 0): 3, (1, 0): 4, (0, 1): 5, (-1, 1): 6, (1, 1): 7}


def move_index(side, usi):
    """class * 81 + destination, both in the mover's orientation.  Classes: the 8 queen
    directions (0..7) and the two knight jumps (8, 9), +10 with promotion, then drops of
    Pawn Lance Knight Silver Bishop Rook Gold (20..26, the core's piece-type order)."""
    to = usi_square(usi[2:4])
    if side == 1:
        to = 80 - to
    if usi[1] == "*":
        return (20 + "PLNSBRG".index(usi[0])) * 81 + to
    frm = usi_square(usi[0:2])
    if side == 1:
        frm = 80 - frm
    df, dr = to // 9 - frm // 9, to % 9 - frm % 9
    if dr == -2 and abs(df) == 1:
        cls = 8 if df < 0 else 9
    else:
        n = max(abs(df), abs(dr))
        assert (df == 0 or dr == 0 or abs(df) == abs(dr)) and n > 0, usi
        cls = _DIRS[(df // n, dr // n)]
    if usi.endswith("+"):
        cls += 10
    return cls * 81 + to


def flip_sfen(sfen):
    """The same position with colours exchanged and the board turned by 180 degrees."""
    parts = sfen.split()
    rows = parts[0].split("/")
    flipped = []
    for row in reversed(rows):
        cells, promo = [], ""
        for ch in row:
            if ch == "+":
                promo = "+"
            else:
                cells.append(promo + ch)
                promo = ""
        flipped.append("".join(c.swapcase() for c in reversed(cells)))
    hand = parts[2] if parts[2] == "-" else parts[2].swapcase()
    if hand != "-":  # sfen lists Black's hand first
        upper = "".join(_hand_tokens(hand, True))
        lower = "".join(_hand_tokens(hand, False))
        hand = upper + lower
    return " ".join(["/".join(flipped), "w" if parts[1] == "b" else "b", hand] + parts[3:])


def _hand_tokens(hand, upper):
    count = ""
    for ch in hand:
        if ch.isdigit():
            count += ch
        else:
            if ch.isupper() == upper:
                yield count + ch
            count = ""


def flip_usi(usi):
    def sq(t):
        return chr(ord("1") + 8 - (ord(t[0]) - ord("1"))) + chr(ord("a") + 8 - (ord(t[1]) - ord("a")))
    if usi[1] == "*":
        return usi[:2] + sq(usi[2:4])
    return sq(usi[0:2]) + sq(usi[2:4]) + usi[4:]


def parse_dump_line(line):
    """One line of `perft features` -> (sfen, bitboards uint64 [86, 2], [(usi, index), ...])."""
    sfen, hexes, moves = line.rstrip("\n").split("\t")
    bb = np.frombuffer(bytes.fromhex(hexes), dtype="<u8").reshape(NUM_PLANES, 2).copy()
    pairs = []
    for tok in moves.split():
        usi, idx = tok.rsplit(":", 1)
        pairs.append((usi, int(idx)))
    return sfen, bb, pairs
