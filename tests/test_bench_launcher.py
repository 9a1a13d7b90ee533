"""bench.py's rank handling (the reference multiplies executors from one command,
/root/reference/src/selfplay/main.cc:33,189-195): `--gpus N` without a launcher starts N ranks itself,
`--gpus` must equal the launcher's WORLD_SIZE, and fewer devices than ranks is an error, never a silent
one-rank run.  The multi-rank skeleton is rehearsed on the CPU with `--executor random` over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _run(args, env=None, timeout=600):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout,
                          env=env or _env())


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--executor", "random", "--steps", "3", "--warmup", "1", "--batch", "8"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["executor"] == "random" and "child process" in d["launched_by"]
    assert d["config"]["weights_broadcast_bytes"] > 0  # rank 1 received rank 0's blob
    assert abs(d["value"] - 8 * 3 * 2 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]  # whole job / max-over-ranks time


def test_under_torch_distributed_run_as_the_driver_launches_it():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", BENCH, "--gpus", "2",
                        "--executor", "random", "--steps", "2", "--warmup", "1", "--batch", "4"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and "launched_by" not in d


def test_gpus_must_match_the_launchers_world_size():
    r = _run(["--gpus", "2", "--executor", "random"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "1", "--executor", "random"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
                                                                MASTER_ADDR="127.0.0.1", MASTER_PORT="29732"))
    assert r.returncode == 2 and not r.stdout.strip()


def test_more_ranks_than_gpus_is_refused():
    """On a machine with fewer than N GPUs `bench.py --gpus N` must fail loudly (here: no GPU at all; on the
    one-GPU box the gpu-marked twin below runs the same command)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this machine has two GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_more_ranks_than_gpus_is_refused_on_the_gpu_box():
    test_more_ranks_than_gpus_is_refused()


def test_cpu_baseline_times_the_product_executor(nsg):
    """bench.py's cpu_baseline leg: the PRODUCT's restatement of the reference's EXECUTOR=random path
    (nsg_cpu_executor, csrc/cpu_executor.cc, the reference's release flags), one core and thread pools; bounded."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline(nsg, seconds=0.3, pool_seconds=0.3, share_threads=2)
    assert out["kind"] == "port" and out["cores"] == 1 and out["value"] > 1000 and "nsg_cpu_executor" in out["sample"]
    assert out["gpu_share_16_threads"]["cores"] <= 2 and out["gpu_share_16_threads"]["value"] > 1000
    assert out["all_cores"]["cores"] == out["all_cores"]["physical_cores_available"] >= 1
    assert out["all_cores"]["value"] > 1000
    # the release-flag build of the executor still produces the known-answer vector G1
    import numpy as np
    ex = nsg.CpuExecutor("random", seed=0)
    p, w, d = ex.compute_blocking(1)
    assert p[0, :4].view(np.uint32).tolist() == [0x3e23a0df, 0x3f7dfd3a, 0x3d221321, 0x3f18f569]
