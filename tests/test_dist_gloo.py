"""N > 1 path on CPU: world_size-2 gloo.  Covers what bench.py does across
ranks: weight-blob broadcast from rank 0 (the only collective), position
sharding with no data-path exchange, and the max-over-ranks timing."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nsg = importlib.import_module("nshogi-engine_amd")
    import oracle_lib
    blob = None
    if rank == 0:
        blob = nsg.weights.to_blob(nsg.weights.make_random(1, 64, seed=5, bn="random"))
    t = nsg.dist.broadcast_blob(blob, src=0, device="cpu")
    got = t.numpy().tobytes()
    # every rank can now build its evaluator from the broadcast bytes; here the
    # oracle stands in for the device (no GPU in this test) on the rank's own shard
    net = oracle_lib.load().net(got)
    bb = nsg.synth.random_batch(3, 86, seed=nsg.dist.shard_seed(7, rank))
    p, v, d = net.evaluate(bb)
    lo, hi = nsg.dist.shard_range(10, rank, world)
    slowest = nsg.dist.max_over_ranks(1.0 + rank)
    total = nsg.dist.sum_over_ranks(hi - lo)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), blob=np.frombuffer(got, dtype=np.uint8),
             policy=p, bb=bb, lo=lo, hi=hi, slowest=slowest, total=total)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path, nsg):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    ref = nsg.weights.to_blob(nsg.weights.make_random(1, 64, seed=5, bn="random"))
    assert r0["blob"].tobytes() == ref and r1["blob"].tobytes() == ref  # broadcast delivered the blob
    assert not np.array_equal(r0["bb"], r1["bb"])      # ranks evaluate distinct positions
    assert not np.array_equal(r0["policy"], r1["policy"])
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 5, 5, 10)
    assert float(r0["slowest"]) == float(r1["slowest"]) == 2.0   # max over ranks
    assert float(r0["total"]) == 10.0


def test_shard_range_properties(nsg):
    for total in (0, 1, 7, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [nsg.dist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
