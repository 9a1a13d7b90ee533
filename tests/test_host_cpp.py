"""The C++ host mirror (infer::Hip / infer::Random... + evaluate::Evaluator over
the C ABI) driven like the engine's evaluation thread
(src/mcts/evaluationworker.cc:124-195)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "nshogi-engine_amd", "csrc", "host")


def _build():
    if not os.path.exists(os.path.join(HOST, "host_eval")):
        import __graft_entry__
        __graft_entry__.build()


def run_host_eval(tmp_path, bb, batch_max, executor, weights="none", precision=0):
    _build()
    f = tmp_path / "features.bin"
    o = tmp_path / "out.bin"
    bb.tofile(str(f))
    r = subprocess.run([os.path.join(HOST, "host_eval"), str(weights), str(f), str(bb.shape[0]),
                        str(batch_max), str(o), str(precision), executor],
                       capture_output=True, text=True, timeout=600)
    out = np.fromfile(str(o), dtype=np.float32).reshape(bb.shape[0], 2189) if r.returncode == 0 else None
    return r, out


def test_cpu_executors_through_cpp_adapter(nsg, oracle, tmp_path):
    bb = nsg.synth.random_batch(23, 86, seed=1)
    r, out = run_host_eval(tmp_path, bb, 5, "random")
    assert r.returncode == 0, r.stderr
    p, w, d = oracle.random_compute(oracle.mt(0), 23)  # seed 0 as both call sites use
    np.testing.assert_array_equal(out[:, :2187], p)
    np.testing.assert_array_equal(out[:, 2187], w)
    np.testing.assert_array_equal(out[:, 2188], d)
    r, out = run_host_eval(tmp_path, bb, 5, "zero")
    assert r.returncode == 0 and not out.any()
    r, out = run_host_eval(tmp_path, bb, 5, "nothing")
    assert r.returncode == 0  # Nothing leaves the Evaluator buffers as they were (nothing.cc:22-24)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [(0, 2e-4), (3, 2e-4)])
def test_hip_through_cpp_adapter(nsg, oracle, tmp_path, precision, tol):
    w = nsg.weights.make_random(2, 64, seed=11, bn="random")
    blob = nsg.weights.to_blob(w)
    path = tmp_path / "net.nsgw"
    path.write_bytes(blob)
    bb = nsg.synth.random_batch(40, 86, seed=12)
    r, out = run_host_eval(tmp_path, bb, 9, "hip", weights=path, precision=precision)
    assert r.returncode == 0, r.stderr
    assert "pinned=1" in r.stdout
    p, v, d = oracle.net(blob).evaluate(bb)
    assert np.abs(out[:, :2187] - p).max() < tol
    assert np.abs(out[:, 2187] - v).max() < tol and np.abs(out[:, 2188] - d).max() < tol


@pytest.mark.gpu
def test_hip_adapter_error_behaviour(nsg, tmp_path):
    """TensorRT::load throws std::runtime_error when the file cannot be opened
    (trt.cc:34-36); the adapter does the same."""
    bb = nsg.synth.random_batch(2, 86)
    r, _ = run_host_eval(tmp_path, bb, 2, "hip", weights=tmp_path / "missing.nsgw")
    assert r.returncode == 3 and "Could not open the file" in r.stderr


@pytest.mark.gpu
def test_batchsize_bench_runs(nsg, tmp_path):
    """The reference's evals/sec harness (bench/batchsize.cc) restated."""
    _build()
    path = tmp_path / "net.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(2, 64, seed=1))
    # no precision argument and no setPrecision call -- what the engine's ladder does
    # (INTEGRATION.md 1): NSG_PRECISION in the environment picks the arithmetic
    for env_prec, want in ((None, 0), ("f16m6", 5), ("3", 3)):
        env = dict(os.environ)
        env.pop("NSG_PRECISION", None)
        if env_prec is not None:
            env["NSG_PRECISION"] = env_prec
        r = subprocess.run([os.path.join(HOST, "batchsize_bench"), str(path), "20", "60", "62"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().split("\n")
        assert f"precision {want}," in lines[1], lines[1]
        rows = [ln.split(",") for ln in lines[2:]]
        assert [int(x[0]) for x in rows] == [60, 61, 62]
        assert all(float(x[2]) > 0 for x in rows)
    env["NSG_PRECISION"] = "tf32"
    r = subprocess.run([os.path.join(HOST, "batchsize_bench"), str(path), "20", "60", "60"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "NSG_PRECISION=tf32" in r.stderr


def test_batchsize_bench_random_executor_initial_position(nsg, tmp_path):
    """EXECUTOR=random through the same harness on the CPU (BASELINE configs[0] plumbing)."""
    _build()
    r = subprocess.run([os.path.join(HOST, "batchsize_bench"), "none", "3", "5", "6", "env", "random"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rows = [ln.split(",") for ln in r.stdout.strip().split("\n")[1:]]
    assert [int(x[0]) for x in rows] == [5, 6]


@pytest.mark.parametrize("cfg", ["5000 4 64 4 2 2", "3000 8 7 3 1 0", "2000 1 512 3 2 4", "8000 6 33 5 3 3"])
def test_batch_pipeline_routing_cpu(cfg):
    """evaluate::BatchPipeline (a9/a10) under concurrency with a checksum executor:
    every leaf is fed exactly once with the outputs computed from its own features."""
    _build()
    r = subprocess.run([os.path.join(HOST, "pipeline_test"), "checksum"] + cfg.split(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "missing 0 dup 0 wrong 0" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["3000 4 64 4 2 2", "1500 3 17 3 1 1"])
def test_batch_pipeline_hip_bit_identical(nsg, tmp_path, cfg):
    """Pipelined (double-buffered, direct-to-pinned) results are bit-identical to the
    plain blocking Evaluator path for every leaf."""
    _build()
    path = tmp_path / "net.nsgw"
    nsg.weights.save(str(path), nsg.weights.make_random(2, 64, seed=5, bn="random"))
    r = subprocess.run([os.path.join(HOST, "pipeline_test"), "hip"] + cfg.split() + [str(path)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "missing 0 dup 0 wrong 0" in r.stdout


def test_evaluator_numa_placement_and_pool_size_check():
    """Evaluator(NumaPlacement=true): thread bound to node ThreadId % nodes, buffers first-touched
    there (evaluator.cc:46-76 restated on /sys + sched_setaffinity); one-node machines are left alone.
    BatchPipeline refuses a buffer pool smaller than Depth + 1 instead of deadlocking in openNext."""
    _build()
    r = subprocess.run([os.path.join(HOST, "pipeline_test"), "numa"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "numa ok" in r.stdout, r.stdout + r.stderr
    assert int(r.stdout.split()[1]) >= 1
    r = subprocess.run([os.path.join(HOST, "pipeline_test"), "badpool"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "refused" in r.stdout, r.stdout + r.stderr
