"""Reader of the self-play teacher files ("NSGT" v1) written by csrc/selfplay/teacher.{h,cc}
(role of the reference's SaveWorker::save, src/selfplay/saveworker.cc:160-182; the record layout is
this build's own, see teacher.h)."""
import numpy as np

RECORD = np.dtype([
    ("board", "u1", (81,)),        # (color << 4) | piece type, 0 = empty; square = file*9 + rank
    ("hands", "u1", (2, 7)),       # [color][pawn, lance, knight, silver, bishop, rook, gold]
    ("side_to_move", "u1"),
    ("winner", "u1"),              # 0 black, 1 white, 2 draw
    ("declare27", "u1"),
    ("ply", "<u2"),
    ("next_move16", "<u2"),        # to[0:7) from[7:14) (81 + type for drops) promote[14]
    ("max_ply", "<u2"),
    ("black_draw_value", "<f4"),
    ("white_draw_value", "<f4"),
    ("game_length", "<u2"),
    ("reserved", "u1", (14,)),
])
assert RECORD.itemsize == 128


def load(path):
    """Returns the records of a teacher file as a numpy structured array."""
    with open(path, "rb") as f:
        head = np.frombuffer(f.read(16), "<u4")
        if head.size != 4 or head[0] != 0x5447534E:
            raise ValueError(f"{path}: not an NSGT teacher file")
        if head[1] != 1 or head[2] != RECORD.itemsize:
            raise ValueError(f"{path}: unsupported NSGT version {head[1]} / record size {head[2]}")
        body = f.read()
    if len(body) % RECORD.itemsize:
        raise ValueError(f"{path}: truncated record ({len(body)} bytes after the header)")
    return np.frombuffer(body, RECORD)


def apply_move(board, hands, side, move16):
    """Plays move16 on (board[81], hands[2][7]) copies: the position the next record must hold."""
    board, hands = board.copy(), hands.copy()
    to, frm, promote = move16 & 127, (move16 >> 7) & 127, (move16 >> 14) & 1
    if frm >= 81:  # drop of piece type frm - 81
        t = frm - 81
        hands[side][t - 1] -= 1
        board[to] = (side << 4) | t
        return board, hands
    piece = int(board[frm])
    if board[to]:
        cap = int(board[to]) & 15
        cap = cap - 8 if cap >= 9 else cap  # captured pieces lose their promotion
        hands[side][cap - 1] += 1
    board[frm] = 0
    board[to] = piece + 8 if promote else piece
    return board, hands
