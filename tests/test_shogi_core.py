"""The shogi rules core (csrc/shogi) and the self-play engine (csrc/selfplay).
libnshogi (the reference's rules library) is absent, so the core is pinned by the
game's public perft numbers, the 593-move maximum position, and a cross-check of the
fast legal-move generator against an independently structured make/unmake generator."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SP = os.path.join(ROOT, "nshogi-engine_amd", "csrc", "selfplay")


def _tool(name):
    path = os.path.join(SP, name)
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    return path


def run(name, *args, timeout=600):
    r = subprocess.run([_tool(name)] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_perft_startpos_known_values():
    # public reference values for shogi perft from the initial position
    out = run("perft", "perft", 5).split()
    assert out == ["1", "30", "2", "900", "3", "25470", "4", "719731", "5", "19861490"]


def test_maximum_legal_moves_position():
    # the well-known position with the maximum of 593 legal moves (the size of MoveList and of Frame's buffers)
    out = run("perft", "movecount", "R8/2K1S1SSk/4B4/9/9/9/9/9/1L1L1L3 b RBGSNLP3g3n17p 1").split()
    assert out[0] == "593" and out[1] == "593"


@pytest.mark.parametrize("sfen,count", [
    # nifu + drop restrictions: black has a pawn in hand; files with a black pawn are excluded
    ("4k4/9/9/9/9/9/P8/9/4K4 b P 1", None),
    # uchifuzume: dropping a pawn in front of the cornered king would be mate -> illegal
    ("8k/7pp/7G1/9/9/9/9/9/K8 b P 1", None),
])
def test_fast_equals_slow_on_rule_positions(sfen, count):
    out = run("perft", "movecount", sfen).split()
    assert out[0] == out[1]


def test_uchifuzume_is_excluded():
    # White king 1a boxed in by its own lance 2a and pawn 2b; Black's gold on 2c guards 1b.
    # P*1b would be checkmate by a dropped pawn -> illegal; every other pawn drop on file 1 is fine.
    sfen = "7lk/7p1/7G1/9/9/9/9/9/K8 b P 1"
    fast, slow = run("perft", "movecount", sfen).split()[:2]
    assert fast == slow
    moves = run("perft", "moves", sfen).split()
    assert "P*1b" not in moves and "P*1c" in moves and "P*5e" in moves
    # with a flight square for the king (no lance on 2a) the same drop is legal
    moves = run("perft", "moves", "8k/7p1/7G1/9/9/9/9/9/K8 b P 1").split()
    assert "P*1b" in moves


def test_generators_agree_on_random_playouts():
    out = run("perft", "crosscheck", 300, 11)
    assert out.startswith("ok positions")


def test_mate_search_known_positions():
    """findMate (role of solver::dfs::solve(State, 3), selfplay/worker.cc:349-358)."""
    # gold drop / gold move mates a cornered king in one
    a, b = run("perft", "mate", 1, "8k/9/7GG/9/9/9/9/9/K8 b - 1").split()
    assert a != "none" and b != "none"
    # the same king with no attackers nearby: no mate within three plies
    assert run("perft", "mate", 3, "8k/9/9/9/9/9/9/9/K7G b - 1").split() == ["none", "none"]
    # a random-playout position (perft matesample 4) with a mate in three but not in one
    sfen = "l2k5/4s2r1/P1p1l3G/1p3g1pP/LbPp5/1N1P1sP1p/+n+r1gP+p1PS/6pBL/+n3SK2G w 2Pn2p 182"
    assert run("perft", "mate", 1, sfen).split() == ["none", "none"]
    assert run("perft", "mate", 3, sfen).split() == ["6g5h", "6g5h"]


def test_mate_search_agrees_with_exhaustive_on_random_playouts():
    """givesCheck vs make-the-move-and-look for every legal move, the check-prefiltered
    search vs the unfiltered one, and every found mate verified against every defence."""
    out = run("perft", "matecheck", 80, 13)
    assert out.strip().endswith("ok") and "mate3" in out


def test_dfpn_solver_is_sound_and_finds_shallow_mates():
    """DfpnSolver (role of libnshogi's solver::dfpn::Solver at selfplay/worker.cc:516-524) on
    random-playout positions: never disproves a position findMate(3) solves, restores the position,
    and every mate it returns is replayed against every defence down to real checkmates."""
    out = run("perft", "dfpncheck", 10, 7, 20000).split()
    assert out[-1] == "ok"
    stats = dict(zip(out[0:-1:2], out[1:-1:2]))
    assert int(stats["dfpn"]) >= int(stats["mate3"]) > 0 and int(stats["deeper"]) > 0
    assert int(stats["verified"]) > 0.9 * int(stats["dfpn"])
    # hand-made cases: a gold dropped in front of the king on a square the pawn guards is mate;
    # without the pawn the bare gold never mates (disproved, not just out of budget)
    assert run("perft", "dfpn", 1000, "4k4/9/4P4/9/9/9/9/9/4K4 b G 1").split()[:1] == ["G*5b"]
    lone = run("perft", "dfpn", 100000, "4k4/9/9/9/9/9/9/9/4K4 b G 1").split()
    assert lone[0] == "none" and int(lone[2]) < 1000
    # the mate-in-three sample of the test above, through the solver
    sfen = run("perft", "matesample", 3).strip()
    a = run("perft", "mate", 3, sfen).split()[0]
    d = run("perft", "dfpn", 100000, sfen).split()
    assert d[0] != "none" and a != "none" and len(d[d.index("pv") + 1:]) >= 3


def test_selfplay_cpu_random_executor_reproducible():
    """EXECUTOR=random self-play (BASELINE config 1 plumbing): finishes games and is
    bit-reproducible under a fixed seed; a different seed plays different games."""
    args = ["--executor", "random", "--threads", "2", "--games-per-group", "4", "--playouts", "30",
            "--max-games", "6"]
    a = json.loads(run("selfplay", *args, "--seed", 5))
    b = json.loads(run("selfplay", *args, "--seed", 5))
    c = json.loads(run("selfplay", *args, "--seed", 6))
    assert a["games_finished"] >= 6 and a["digest"] == b["digest"] and a["moves"] == b["moves"]
    assert c["digest"] != a["digest"]
    assert a["black"] + a["white"] + a["draw"] == a["games_finished"]


def test_selfplay_teacher_records(nsg, tmp_path):
    """--teacher (role of SaveWorker::save, saveworker.cc:160-182): one record per full-search
    ply of every finished game.  Checked here: the count reported, 40 pieces in every position,
    ply parity, consistent game-level fields, and -- with every search a full search -- that
    applying a record's move yields exactly the next record's position, from the initial position
    to the end of each game."""
    path = str(tmp_path / "t.nsgt")
    out = json.loads(run("selfplay", "--executor", "random", "--threads", "2", "--games-per-group", "3",
                         "--playouts", "24", "--max-games", "4", "--seed", 11, "--full-search-ratio", "1.0",
                         "--teacher", path))
    rec = nsg.teacher.load(path)
    assert len(rec) == out["teacher_records"] > 0
    assert np.all(rec["winner"] <= 2) and np.all(rec["side_to_move"] == rec["ply"] % 2)
    on_board = (rec["board"] != 0).sum(axis=1) + rec["hands"].reshape(len(rec), -1).sum(axis=1)
    assert np.all(on_board == 40)
    assert np.all(((rec["board"] & 15) == 8).sum(axis=1) == 2)  # both kings
    # games are written whole and in ply order: a new game starts where the ply does not grow.
    # Positions with a single legal reply are never "full" searches (worker.cc:171-176), so a few
    # plies may be missing; everything else must be there at ratio 1.0.
    cuts = [0] + [i for i in range(1, len(rec)) if rec["ply"][i] <= rec["ply"][i - 1]] + [len(rec)]
    assert len(cuts) - 1 == out["games_finished"]
    assert len(rec) > 0.9 * sum(int(rec["game_length"][c]) for c in cuts[:-1])
    chained = 0
    for s, e in zip(cuts[:-1], cuts[1:]):
        g = rec[s:e]
        assert g["ply"][0] == 0 and np.array_equal(g["board"][0], rec["board"][0])  # all from the initial position
        assert len(set(g["winner"])) == 1 and len(set(g["max_ply"])) == 1 and len(set(g["game_length"])) == 1
        assert np.all(g["ply"] < g["game_length"])
        assert np.all(g["black_draw_value"] + g["white_draw_value"] == np.float32(1.0))
        for k in range(len(g) - 1):
            if g["ply"][k + 1] != g["ply"][k] + 1:
                continue
            b, h = nsg.teacher.apply_move(g["board"][k], g["hands"][k], int(g["side_to_move"][k]), int(g["next_move16"][k]))
            assert np.array_equal(b, g["board"][k + 1]) and np.array_equal(h, g["hands"][k + 1])
            chained += 1
    assert chained > 0.8 * len(rec)
    # the default ratio writes only a subset
    out2 = json.loads(run("selfplay", "--executor", "random", "--threads", "1", "--games-per-group", "3",
                          "--playouts", "24", "--max-games", "2", "--seed", 11, "--teacher", path))
    rec2 = nsg.teacher.load(path)
    assert 0 < len(rec2) == out2["teacher_records"] < out2["moves"]


def test_selfplay_gumbel_mode_cpu():
    """--gumbel (Gumbel AlphaZero root with sequential halving, worker.cc:428-475,784-905)."""
    args = ["--executor", "random", "--threads", "1", "--games-per-group", "4", "--playouts", "64",
            "--num-sampling-moves", "16", "--gumbel", "1", "--max-games", "3"]
    a = json.loads(run("selfplay", *args, "--seed", 3))
    b = json.loads(run("selfplay", *args, "--seed", 3))
    assert a["games_finished"] >= 3 and a["digest"] == b["digest"]
    assert a["playouts_per_sec"] > 0 and a["moves"] > 0


def _game_log(path):
    games = {}
    for ln in open(path):
        gid, rest = ln.split(" ", 1)
        games[int(gid)] = rest
    return games


def test_selfplay_games_do_not_depend_on_grouping(tmp_path):
    """Game slot s plays the games s, s+N, s+2N, ... each from its own RNG stream, so with an
    executor whose outputs do not depend on batch composition (infer::Zero here) every game is the
    same game however the N slots are spread: 1 thread x 2 groups x 6, 2 threads x 2 x 3 sharing one
    evaluation cache, or two GPU shards (--num-gpus 2; the CPU executors stand in for the devices)
    x 1 thread x 2 x 3, one engine whose games are advanced by three host threads (--workers 3), or with
    the df-pn call of judge handed to a pool of solver threads (--solver-threads 2)."""
    base = ["--executor", "zero", "--playouts", "40", "--seed", "3", "--dfpn-nodes", "2000", "--max-games", "10"]
    logs = []
    for i, shape in enumerate((["--threads", "1", "--games-per-group", "6"],
                               ["--threads", "2", "--games-per-group", "3", "--share-evaluation-cache", "1"],
                               ["--num-gpus", "2", "--threads", "1", "--games-per-group", "3"],
                               ["--threads", "1", "--workers", "3", "--games-per-group", "6"],
                               ["--threads", "1", "--workers", "2", "--solver-threads", "2", "--games-per-group", "6"])):
        path = tmp_path / f"g{i}.log"
        out = json.loads(run("selfplay", *base, *shape, "--game-log", path))
        assert out["concurrent_games"] == 12 and out["games_finished"] >= 10
        logs.append(_game_log(path))
    assert json.loads(run("selfplay", *base, "--num-gpus", "2", "--threads", "1", "--games-per-group", "3"))["num_gpus"] == 2
    common = set(logs[0]) & set(logs[1]) & set(logs[2]) & set(logs[3]) & set(logs[4])
    assert len(common) >= 6
    for gid in common:
        assert logs[0][gid] == logs[1][gid] == logs[2][gid] == logs[3][gid] == logs[4][gid], gid
    # the per-GPU shards both worked
    out = json.loads(run("selfplay", *base, "--num-gpus", "2", "--threads", "2", "--games-per-group", "2"))
    assert len(out["evals_per_sec_by_gpu"]) == 2 and all(x > 0 for x in out["evals_per_sec_by_gpu"])


def test_selfplay_windowed_rate_and_cache_options():
    out = json.loads(run("selfplay", "--executor", "random", "--threads", "1", "--games-per-group", "4", "--playouts", "16",
                         "--seconds", "3", "--seed", "2", "--dfpn-nodes", "0", "--mate-search", "0",
                         "--evaluation-cache-memory-size", "16"))
    assert out["games_finished"] > 0 and out["games_per_sec_window"] > 0 and 1.0 < out["window_seconds"] < 2.5
    assert out["evaluation_cache_mb_per_gpu"] == 16 and 0 < out["cache_hit_ratio"] < 1
    off = json.loads(run("selfplay", "--executor", "random", "--threads", "1", "--games-per-group", "4", "--playouts", "16",
                         "--max-games", "2", "--seed", "2", "--dfpn-nodes", "0", "--evaluation-cache-memory-size", "0"))
    assert off["cache_hit_ratio"] == 0


@pytest.mark.gpu
def test_selfplay_hip_games_do_not_depend_on_grouping(nsg, tmp_path, monkeypatch):
    """The same on the HIP evaluator, in the arithmetic that does not depend on the batch size: the
    f32 path with one fixed tile plan (NSG_CONV_NB=1, NSG_CONV_NFRAG=4 -- by default the plan, and
    with it the f32 summation order, follows the batch size).  The evaluation cache is off: like the
    reference's it is keyed by the position alone while the planes also carry the ply and the game's
    StateConfig, so a hit may hand a game another game's evaluation of the same board."""
    monkeypatch.setenv("NSG_CONV_NB", "1")
    monkeypatch.setenv("NSG_CONV_NFRAG", "4")
    path = tmp_path / "net.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(2, 64, seed=3, bn="random"))
    base = ["--executor", "hip", "--weights", str(path), "--precision", "0", "--playouts", "24", "--seed", "9",
            "--max-games", "8", "--dfpn-nodes", "2000", "--evaluation-cache-memory-size", "0"]
    logs = []
    for i, shape in enumerate((["--threads", "1", "--games-per-group", "6"], ["--threads", "2", "--games-per-group", "3"])):
        lp = tmp_path / f"h{i}.log"
        json.loads(run("selfplay", *base, *shape, "--game-log", lp))
        logs.append(_game_log(lp))
    common = set(logs[0]) & set(logs[1])
    assert len(common) >= 4
    for gid in common:
        assert logs[0][gid] == logs[1][gid], gid


@pytest.mark.gpu
def test_selfplay_two_gpu_shards_on_the_one_device(nsg, tmp_path, monkeypatch):
    """`selfplay --num-gpus 2` (selfplay/main.cc:33,189-195) on the one-GPU box: both shards mapped to device 0
    (--gpu-map 0,0) and NSG_SHARED_FORCE_COPY=1, so the first executor of shard 1 takes nsg_load_shared's
    OTHER-device branch (own allocations + hipMemcpyPeer of every packed layer) and the shards' engine threads
    bind their device themselves.  Same games as the one-shard run: the copied weights are the same weights."""
    monkeypatch.setenv("NSG_CONV_NB", "1")
    monkeypatch.setenv("NSG_CONV_NFRAG", "4")
    path = tmp_path / "net.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(2, 64, seed=3, bn="random"))
    base = ["--executor", "hip", "--weights", str(path), "--precision", "0", "--playouts", "24", "--seed", "9",
            "--max-games", "8", "--dfpn-nodes", "2000", "--evaluation-cache-memory-size", "0"]
    lp0, lp1 = tmp_path / "one.log", tmp_path / "two.log"
    one = json.loads(run("selfplay", *base, "--threads", "2", "--games-per-group", "3", "--game-log", lp0))
    monkeypatch.setenv("NSG_SHARED_FORCE_COPY", "1")
    two = json.loads(run("selfplay", *base, "--num-gpus", "2", "--gpu-map", "0,0", "--threads", "1",
                         "--games-per-group", "3", "--game-log", lp1))
    assert one["num_gpus"] == 1 and two["num_gpus"] == 2 and two["concurrent_games"] == one["concurrent_games"] == 12
    assert len(two["evals_per_sec_by_gpu"]) == 2 and all(x > 0 for x in two["evals_per_sec_by_gpu"])
    a, b = _game_log(lp0), _game_log(lp1)
    common = set(a) & set(b)
    assert len(common) >= 4
    for gid in common:
        assert a[gid] == b[gid], gid


@pytest.mark.gpu
def test_selfplay_hip_reproducible(nsg, tmp_path):
    """Self-play on the HIP evaluator: move selection is bit-identical under a fixed seed
    (north_star) from run to run of one configuration (independence from the grouping of the game
    slots: test_selfplay_hip_games_do_not_depend_on_grouping)."""
    path = tmp_path / "net.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(2, 64, seed=3, bn="random"))
    base = ["--executor", "hip", "--weights", str(path), "--playouts", "24", "--max-games", "4", "--seed", "9",
            "--threads", "1", "--games-per-group", "6"]
    a = json.loads(run("selfplay", *base))
    b = json.loads(run("selfplay", *base))
    assert a["games_finished"] >= 4 and a["digest"] == b["digest"] and a["moves"] == b["moves"]
    assert a["evals_per_sec"] > 0 and 0 <= a["cache_hit_ratio"] < 1


@pytest.mark.gpu
def test_selfplay_gumbel_mode_hip(nsg, tmp_path):
    """--gumbel on the HIP evaluator (Gumbel AlphaZero root with sequential halving, worker.cc:428-475,784-905): the
    policy logits the halving ranks now come from the network, not from the uniform stand-in executor.  Reproducible
    from the seed, independent of the grouping of the game slots (a game's arithmetic depends on the batch's tile
    plan only through bit-identical kernels: 64 channels, per-layer kernels), and different from the AlphaZero-mode
    games of the same seed (the root selection rule changed, not just the noise)."""
    path = tmp_path / "net.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(2, 64, seed=3, bn="random"))
    base = ["--executor", "hip", "--weights", str(path), "--playouts", "32", "--num-sampling-moves", "16",
            "--max-games", "4", "--seed", "9", "--threads", "1"]
    la, lb, lc = tmp_path / "a.log", tmp_path / "b.log", tmp_path / "c.log"
    a = json.loads(run("selfplay", *base, "--gumbel", "1", "--games-per-group", "6", "--game-log", la))
    b = json.loads(run("selfplay", *base, "--gumbel", "1", "--games-per-group", "6", "--game-log", lb))
    assert a["games_finished"] >= 4 and a["digest"] == b["digest"] and a["moves"] == b["moves"]
    assert a["evals_per_sec"] > 0 and a["playouts_per_sec"] > 0
    c = json.loads(run("selfplay", *base, "--gumbel", "0", "--games-per-group", "6", "--game-log", lc))
    ga, gc = _game_log(la), _game_log(lc)
    common = set(ga) & set(gc)
    assert common and any(ga[g] != gc[g] for g in common)


@pytest.mark.gpu
def test_selfplay_team_trunk_two_engine_threads_reproducible(nsg, tmp_path):
    """Small leaf batches of a 256-channel net run the team trunk (one persistent launch per forward, one such launch
    per device at a time).  Two engine threads = four evaluators hand the device's token to each other all the time:
    the run must neither time out (the token is handed over under a lock that is held until the new holder's launch
    is in its stream -- released earlier, two launches once ran side by side and starved each other) nor depend on
    the interleaving (every game the same from run to run)."""
    path = tmp_path / "net.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(2, 256, seed=5, bn="random"))
    base = ["--executor", "hip", "--weights", str(path), "--precision", "5", "--playouts", "32", "--seed", "11",
            "--max-games", "8", "--threads", "2", "--games-per-group", "3", "--evaluation-cache-memory-size", "0"]
    la, lb = tmp_path / "a.log", tmp_path / "b.log"
    a = json.loads(run("selfplay", *base, "--game-log", la))
    b = json.loads(run("selfplay", *base, "--game-log", lb))
    assert a["games_finished"] >= 8 and b["games_finished"] >= 8 and a["avg_batch"] <= 3
    ga, gb = _game_log(la), _game_log(lb)
    common = set(ga) & set(gb)
    assert len(common) >= 6
    for gid in common:
        assert ga[gid] == gb[gid], gid


def test_selfplay_packed_batches_against_the_sfen_restatement(tmp_path):
    """Host batch packing of the self-play path (selfplay::EvaluationWorker::doTask):
    for EVERY leaf of every batch, the 1376 bytes the engine packed into that slot of the pinned batch
    buffer expand to the planes tests/shogi_ref.py writes from the leaf's SFEN and StateConfig alone
    (MaxPly and the draw values differ from game to game, worker.cc:132-150); slots are 0..N-1 without
    gaps, also with several host workers packing their own slot ranges."""
    import shogi_ref
    import oracle_lib
    orc = oracle_lib.load()
    for workers in (1, 3):
        path = tmp_path / f"leaves{workers}.txt"
        run("selfplay", "--executor", "hash", "--threads", "1", "--workers", workers, "--games-per-group", "7",
            "--playouts", "40", "--max-games", "3", "--seed", "6", "--dfpn-nodes", "2000", "--leaf-log", path)
        rows = [ln.rstrip("\n").split("\t") for ln in open(path)]
        assert len(rows) > 3000
        batches = {}
        for grp, batch, slot, sfen, max_ply, black_draw, hexes in rows:
            batches.setdefault((grp, batch), []).append(int(slot))
        assert all(sorted(v) == list(range(len(v))) for v in batches.values())  # contiguous slots per batch
        assert max(len(v) for v in batches.values()) == 7
        pick = np.random.default_rng(workers).choice(len(rows), size=600, replace=False)
        sample = [rows[i] for i in sorted(pick)]
        bb = np.stack([np.frombuffer(bytes.fromhex(r[6]), dtype="<u8").reshape(86, 2) for r in sample])
        got = orc.extract_bits(bb, True)
        configs = set()
        for r, planes in zip(sample, got):
            want = shogi_ref.expected_planes(r[3], int(r[4]), float(r[5]))
            np.testing.assert_array_equal(planes, want, err_msg=r[3])
            configs.add((r[4], r[5]))
        assert len(configs) > 3  # per-game StateConfigs really differ


def test_selfplay_routing_with_a_position_dependent_executor(tmp_path):
    """--executor hash returns, per position, outputs derived from a checksum of that position's own
    feature bitboards.  A leaf that received another slot's outputs would play a different game: the
    per-game logs must be the same for every way of spreading the game slots (the routing half of a10
    and of feedResult / Frame::setEvaluation, a11)."""
    base = ["--executor", "hash", "--playouts", "48", "--seed", "12", "--dfpn-nodes", "2000", "--max-games", "8"]
    logs = []
    for i, shape in enumerate((["--threads", "1", "--games-per-group", "6"],
                               ["--threads", "3", "--games-per-group", "2"],
                               ["--num-gpus", "2", "--threads", "1", "--workers", "2", "--games-per-group", "3"],
                               ["--threads", "1", "--workers", "3", "--solver-threads", "2", "--games-per-group", "6"])):
        path = tmp_path / f"r{i}.log"
        # (no evaluation cache: like the reference's, it is keyed by the position alone, while the planes also
        # carry the ply and the game's StateConfig -- a hit may return another game's evaluation of the board)
        out = json.loads(run("selfplay", *base, *shape, "--game-log", path, "--evaluation-cache-memory-size", "0"))
        assert out["concurrent_games"] == 12
        logs.append(_game_log(path))
    common = set(logs[0]) & set(logs[1]) & set(logs[2]) & set(logs[3])
    assert len(common) >= 5
    for gid in common:
        assert logs[0][gid] == logs[1][gid] == logs[2][gid] == logs[3][gid], gid
    assert len({logs[0][g] for g in common}) == len(common)  # and the games differ from one another
