"""Multi-GPU plumbing for the evaluator: one process per GPU,
positions sharded across ranks with NO data-path collective; the only
collective is one broadcast of the weight blob at start-up (RCCL over xGMI on
the GPU box, gloo in the CPU tests), replacing every executor re-reading the
model file (/root/reference/src/infer/trt.cc:109-186,
src/mcts/evaluationworker.cc:83-86)."""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_seed(base_seed, rank):
    """Every rank evaluates its own distinct positions (weak scaling)."""
    return int(base_seed) + 1000003 * int(rank)


def shard_range(total, rank, world):
    """Contiguous split of `total` units over `world` ranks (sizes differ by <= 1)."""
    q, r = divmod(int(total), int(world))
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def broadcast_blob(blob, src=0, device="cpu"):
    """Rank `src` passes the weight blob (bytes); every rank gets a uint8 tensor
    on `device` holding it.  Two collectives: the size, then the bytes."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    size = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        size[0] = len(blob)
    dist.broadcast(size, src=src)
    buf = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    if rank == src:
        buf.copy_(torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()))
    dist.broadcast(buf, src=src)
    return buf


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
