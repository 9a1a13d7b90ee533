#!/usr/bin/env python3
"""bench.py -- NN evaluations/sec of the MI355X-native evaluator.

Metric (BASELINE.json): "NN evals/sec at batch=512" on the 20-block x 256-channel
resnet (configs[2]).  Input = the reference benchmark's own: the feature stack of the
initial position in every batch slot (/root/reference/src/bench/batchsize.cc:47-59), built
by this build's rules core.  Two rates are printed, and the line says which is which:
  * `value` -- the measurement contract's device-resident rate: the feature bitboards are
    already in HBM when the timed region starts and the outputs stay in HBM; exactly
    --steps passes are timed (`sustained_evals_per_sec` is the same loop held for >= 5 s; that leg runs
    first, so the --warmup / --steps passes run on a chip already at its steady clocks).
  * `reference_metric.evals_per_sec` -- the reference's own definition,
    batchsize.cc:61-79: BatchSize * Repeat / wall over back-to-back computeBlocking calls
    INCLUDING H2D and D2H, held for >= 5 s; `..._distinct_positions` repeats both on B
    distinct positions of random-playout games.

A "step" = one pass of the hot path over one batch: planes -> resnet -> heads.
One process per GPU; positions shard across ranks with no data-path collective
(weak scaling); RCCL is used once, to broadcast the weight blob from rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (what the driver runs)
    python bench.py --gpus N ...     without a launcher: this process starts the N ranks itself
                                     (torch.distributed.run as a CHILD process) and relays their line;
                                     it fails loudly when the machine has fewer than N GPUs
`--gpus` must equal WORLD_SIZE when a launcher set one; the line carries `ranks_seen` (an all-reduce of 1).
`--executor random` is the CPU rehearsal of the multi-rank path (gloo, the reference's EXECUTOR=random
stand-in as the executor, no GPU touched): tests/test_bench_launcher.py drives it with 2 ranks.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Default precision f16m6: the fastest path that meets the north_star's 1e-3 parity bar
# (measured 1.5e-4 vs the CPU oracle on this net, profiles/r02/b_accuracy_all_precisions_b512.jsonl;
# f16m8 1.5e-4; f16x3 and f32 measure 6e-6 and are reported beside it; plain f16 / bf16 do not meet
# the bar).
# MI355X_MICROARCH.md "Chip-level parameters"
PEAK_TFLOPS = {"fp32": 157.3, "fp16": 2516.6, "bf16": 2516.6, "f16x3": 2516.6, "f16m8": 2516.6, "f16m6": 2516.6}
DTYPE_NAME = {"fp32": "f32", "fp16": "f16", "bf16": "bf16", "f16x3": "f16x3 (split f16 hi/lo, f32 accumulate)",
              "f16m8": "f16m8 (f16 main term + fp8 MX correction terms, f32 accumulate)",
              "f16m6": "f16m6 (f16 main term + e2m3 MX correction terms with per-32-channel E8M0 scales, f32 accumulate)"}
# matrix-pipe work per algorithmic MAC in units of one f16 MFMA MAC (the MX instruction
# retires 4x the K of the f16 one in 2x its cycles; per tap and chunk pair 2 f16 + 1 MX slab)
# (f16m6: the e2m3 form of the MX instruction retires that K in ONE f16 MFMA's cycles: 2 f16 + 1 MX slab = 1.5)
MFMA_UNITS = {"fp32": 1, "fp16": 1, "bf16": 1, "f16x3": 3, "f16m8": 2.0, "f16m6": 1.5}


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def _physical_cores_available():
    """Physical cores this process may run on: distinct (package, core) pairs among its affinity mask."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1
    seen = set()
    for c in cpus:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            seen.add(("?", str(c)))
    return max(1, len(seen))


def cpu_baseline(nsg, seconds=8.0, pool_seconds=5.0, share_threads=16):
    """The reference's EXECUTOR=random CPU path (src/infer/random.cc:28-42 driven like src/bench/batchsize.cc):
    the PRODUCT's restatement of it -- nsg_cpu_executor (csrc/cpu_executor.cc), built with the reference's own release
    flags (Makefile:30-32 + the AVX2 set) -- at batch 512, bounded to ~`seconds` of CPU work per leg:
    (i) one core: what one reference evaluation thread does; (ii) SURVEY.md 8d (ii): one executor + output buffers
    per thread, on the 16 threads that are a one-GPU box's CPU share, and on every physical core this process may
    run on.  ("port": the reference itself cannot be built here -- libnshogi is absent.)"""
    import threading
    B = 512
    model = _cpu_model()

    def one(seed):
        return (nsg.CpuExecutor("random", seed=seed), np.empty((B, 2187), np.float32), np.empty(B, np.float32),
                np.empty(B, np.float32))

    ex, pol, win, drw = one(0)
    ex.compute_blocking(B, policy=pol, win=win, draw=drw)  # warm-up
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ex.compute_blocking(B, policy=pol, win=win, draw=drw)
        n += B
    dt = time.perf_counter() - t0
    built = "nsg_cpu_executor (the product's infer::Random, -O3 -ffast-math + the reference's release flags)"
    out = {"value": n / dt, "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": f"{built}, seed 0, batch {B}, {n} positions in {dt:.1f} s on 1 of {os.cpu_count()} host threads ({model})"}
    phys = _physical_cores_available()

    def pool(T):
        counts = [0] * T
        sets = [one(i) for i in range(T)]
        stop = time.perf_counter() + pool_seconds

        def work(i):  # ctypes releases the GIL inside the call: the threads run in parallel
            e, p, w, d = sets[i]
            while time.perf_counter() < stop:
                e.compute_blocking(B, policy=p, win=w, draw=d)
                counts[i] += B

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        return {"value": sum(counts) / dt, "unit": "evals/s", "cores": T, "kind": "port",
                "sample": f"{T} threads, one {built} (seed i) + output buffers each, batch {B}, {sum(counts)} positions "
                          f"in {dt:.1f} s ({model})"}

    out["gpu_share_16_threads"] = pool(max(1, min(phys, share_threads)))
    out["all_cores"] = pool(phys)
    out["all_cores"]["physical_cores_available"] = phys
    out["all_cores"]["host_threads_total"] = os.cpu_count()
    return out


def pmc_summary(args, B):
    """The committed rocprofv3 PMC passes of this same workload (scripts/pmc.sh ->
    profiles/rNN/pmc_<precision>_summary.json), newest round first.  These are NOT measured by
    this run: PMC collection needs its own rocprofv3 passes; the bench line names the file."""
    if args.net != "20x256" or B != 512:
        return None, None
    for rnd in ("r04", "r03", "r02", "r01"):
        rel = os.path.join("profiles", rnd, f"pmc_{args.precision}_summary.json")
        try:
            return json.load(open(os.path.join(ROOT, rel))), rel
        except (OSError, ValueError):
            continue
    return None, None


def _conv_rows(d):
    # full-tile trunk conv: tileKernel<PREC, MODE 0 conv, 2 boards, 4 fragments, 4 waves, residual?>
    rows = [c for name, c in (d or {}).items() if "trunkKernel" in name]
    return rows or [c for name, c in (d or {}).items() if "tileKernel" in name and ", 0, 2, 4, 4," in name]


def conv_traffic_bytes(d):
    """HBM bytes per trunk-conv launch (FETCH_SIZE doubled for 16-byte-per-lane streams per
    MI355X_MICROARCH.md 'HBM'; counters are in KiB).  Mean over launches with / without residual."""
    v = [(2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 for c in _conv_rows(d)
         if "FETCH_SIZE" in c and "WRITE_SIZE" in c]
    return sum(v) / len(v) if v else None


def conv_mfma_busy(d):
    """Fraction of the trunk conv's run time its matrix pipes were busy:
    SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES) -- one wave per SIMD, SQ_WAVE_CYCLES counts
    quad-cycles."""
    v = [c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"]) for c in _conv_rows(d)
         if c.get("SQ_WAVE_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c]
    return sum(v) / len(v) if v else None


def extract_traffic(B, kernel="extractNCHW"):
    """HBM bytes per extractNCHW launch from the committed PMC passes (scripts/pmc_extract.sh), newest round
    first: (2 x FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per the guide's gfx950 correction."""
    if B != 512:
        return None, None
    for rnd in ("r04", "r03"):
        rel = os.path.join("profiles", rnd, "pmc_extract_summary.json")
        try:
            d = json.load(open(os.path.join(ROOT, rel)))
        except (OSError, ValueError):
            continue
        for name, c in d.items():
            if kernel in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, rel
    return None, None


def extract_roofline(nsg, ev, bb, B, precision):
    """Second roofline entry: plane expansion, HBM-bound.  The kernel the forward pass RUNS
    (extractAct<precision>: the bit selection of extractbit.cu:19-37 written straight into the trunk's input layout,
    [81][padded channels] rows in the evaluator's element format) is timed on the evaluator's own stream
    (nsg_time_planes, HIP events) and is the entry's achieved / frac; the reference-layout kernel behind
    nsg_extract_bits (fp32 NCHW, the reference's K1; not on the forward path) is reported beside it."""
    import torch
    C = 86
    ms_act, bytes_act = ev.time_planes(B, 200)
    src = torch.from_numpy(bb.view(np.int64).copy()).cuda()
    dst = torch.empty(B * C * 81, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for cf, name in ((True, "NCHW"), (False, "NHWC")):
        for _ in range(5):
            nsg.extract_bits(dst.data_ptr(), src.data_ptr(), B, C, cf, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        iters = 200
        e0.record()
        for _ in range(iters):
            nsg.extract_bits(dst.data_ptr(), src.data_ptr(), B, C, cf, stream)
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / iters
    nbytes = B * (1376 + 27864)
    kname = {"f16m6": "extractActM6", "f16m8": "extractActM8"}.get(precision, "extractAct")
    traffic, traffic_file = extract_traffic(B, kname)
    t2, t2_file = extract_traffic(B)
    return {"bound": "hbm", "achieved": bytes_act / ms_act / 1e6, "peak": 8000.0, "unit": "GB/s",
            "frac": bytes_act / ms_act / 1e6 / 8000.0, "traffic": traffic,
            "traffic_source": (f"{traffic_file} (committed rocprofv3 --pmc passes of this kernel at this batch, NOT "
                               f"measured by this run)" if traffic_file else None),
            "kernel": f"{kname} (the plane expansion on the forward path: bitboards -> [81][128] trunk-input rows, "
                      f"{precision} element format, 4 B per channel)",
            "avg_launch_ms": ms_act, "launches_timed": 200, "algorithmic_bytes_per_launch": bytes_act,
            "reference_layout_kernel": {
                "kernel": "extractNCHW (nsg_extract_bits, fp32 channels first: the reference's K1, extractbit.cu:15-39; "
                          "not on the forward path)",
                "achieved": nbytes / out["NCHW"] / 1e6, "frac": nbytes / out["NCHW"] / 1e6 / 8000.0, "unit": "GB/s",
                "avg_launch_ms": out["NCHW"], "algorithmic_bytes_per_launch": nbytes, "traffic": t2,
                "traffic_source": t2_file, "nhwc_avg_launch_ms": out["NHWC"], "nhwc_GB_per_s": nbytes / out["NHWC"] / 1e6},
            "note": "one launch moves 15-22 MB at batch 512: launch + first-byte latency, not bandwidth, sets its time "
                    "(profiles/r02/extract_roofline.json has the batch sweep); 0.5 % of a forward"}


def config_leg(nsg, gpu, net, batch, precision, seconds=1.5):
    """One more BASELINE config on this GPU: device-resident evals/s, the trunk conv's average launch time (HIP
    events on the evaluator's stream) and the fraction of the f16/bf16 MFMA peak, for `net` at `batch`."""
    import torch
    blocks, channels = (int(x) for x in net.split("x"))
    ev = nsg.Evaluator(gpu, batch, 86, precision=precision)
    t0 = time.perf_counter()
    ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(blocks, channels, seed=0, bn="identity")))
    load_s = time.perf_counter() - t0
    ev.upload_features(nsg.positions.startpos_batch(batch))
    for _ in range(3):
        ev.forward_resident(batch)
    torch.cuda.synchronize()
    ev.profile_enable(True)
    ev.profile_read()
    rate, n, took = held_rate(lambda: ev.forward_resident(batch), batch, torch.cuda.synchronize, seconds)
    prof = ev.profile_read()
    ev.profile_enable(False)
    info, plan = ev.info(), ev.last_plan()
    conv_ms = prof["trunk_ms_total"] / max(prof["trunk_launches"], 1)
    one_launch = prof["trunk_launches"] == prof["forwards"]  # (team trunk / persistent trunk: the events bracket ONE launch)
    conv_flops = info["trunk_conv_flops_per_position"] * batch * (2 * blocks if one_launch else 1)
    peak = PEAK_TFLOPS[precision]
    ev.close()
    return {"net": net, "batch": batch, "precision": precision, "trunk_precision": plan["trunk_precision"],
            "evals_per_sec": rate, "whole_net_frac_of_peak": rate * info["flops_per_position"] / 1e12 / peak,
            "conv_avg_launch_ms": conv_ms, "conv_launches_timed": prof["trunk_launches"],
            "conv_frac_of_peak": conv_flops / (conv_ms * 1e-3) / 1e12 / peak if conv_ms > 0 else None,
            "conv_timing_covers": "the one persistent launch of all 3x3 layers" if one_launch else "one F->F 3x3 conv launch",
            "plan": plan, "forwards_timed": n, "seconds": took, "load_seconds_incl_weight_packing": load_s,
            "positions": "startpos"}


def held_rate(fn, B, sync, seconds):
    """evals/s of `fn` called back to back for at least `seconds` (after one untimed call)."""
    fn()
    sync()
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(8):
            fn()
        n += 8
        if time.perf_counter() - t0 >= seconds:
            break
    sync()
    dt = time.perf_counter() - t0
    return B * n / dt, n, dt


def quick_rate(nsg, local_rank, blob, bb, B, precision, steps=5):
    """Short device-resident pass of another precision, for context in the same line."""
    import torch
    ev = nsg.Evaluator(local_rank, B, 86, precision=precision)
    ev.load_memory(blob)
    ev.upload_features(bb)
    ev.forward_resident(B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev.forward_resident(B)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev.close()
    return B * steps / dt


SELFPLAY_BIN = os.path.join(ROOT, "nshogi-engine_amd", "csrc", "selfplay", "selfplay")


def selfplay_leg(weights_path, gpu, seconds, threads, precision, playouts=800, games_per_group=256, workers=1, solvers=0):
    """BASELINE metric #2 on this rank's GPU: the self-play driver (csrc/selfplay) with
    the reference's option names/values of config 4 (--num-playouts 800, batch = games per
    group).  games/sec = finished games / elapsed (saveworker.cc:135-137).  Never raises: a rank
    whose driver fails or hangs must still reach the collectives that follow."""
    import subprocess
    prec = {"fp32": 0, "fp16": 1, "bf16": 2, "f16x3": 3, "f16m8": 4, "f16m6": 5}[precision]
    try:
        r = subprocess.run([SELFPLAY_BIN, "--executor", "hip", "--weights", weights_path, "--gpu", str(gpu),
                            "--threads", str(threads), "--workers", str(workers), "--solver-threads", str(solvers),
                            "--games-per-group", str(games_per_group),
                            "--playouts", str(playouts), "--seconds", str(seconds), "--seed", "1",
                            "--precision", str(prec)], capture_output=True, text=True, timeout=seconds * 3 + 300)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-300:]}
        return json.loads(r.stdout.strip().split("\n")[-1])
    except (subprocess.TimeoutExpired, OSError, ValueError, IndexError) as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def selfplay_cpu_baseline(seconds=8.0):
    """BASELINE configs[0]: EXECUTOR=random CPU path, 1 MCTS thread, 100 playouts, startpos."""
    import subprocess
    r = subprocess.run([SELFPLAY_BIN, "--executor", "random", "--threads", "1", "--games-per-group", "1",
                        "--playouts", "100", "--seconds", str(seconds), "--seed", "1"],
                       capture_output=True, text=True, timeout=seconds * 3 + 60)
    return json.loads(r.stdout.strip().split("\n")[-1]) if r.returncode == 0 else None


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) under
    torch.distributed.run as a CHILD process, relay the one JSON line rank 0 prints, return the child's
    exit code.  This process never initialises the GPU (torch.cuda.device_count() only counts)."""
    import subprocess
    if args.executor == "hip":
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this machine has {have} GPU(s): refusing to report a "
                  f"{args.gpus}-GPU number from fewer devices", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout[-2000:])
        print(f"bench.py: the {args.gpus}-rank run failed (exit {r.returncode})", file=sys.stderr)
        return r.returncode or 1
    line = json.loads(lines[-1])
    if line.get("ranks_seen") != args.gpus or line.get("n_gpus") != args.gpus:
        print(f"bench.py: asked for {args.gpus} ranks, the line reports n_gpus={line.get('n_gpus')} "
              f"ranks_seen={line.get('ranks_seen')}", file=sys.stderr)
        return 3
    line["launched_by"] = "bench.py itself (torch.distributed.run as a child process)"
    print(json.dumps(line))
    return 0


def rehearse_cpu(args, rank, world):
    """--executor random: the multi-rank skeleton of this file on the CPU -- gloo process group, weight-blob
    broadcast from rank 0, per-rank executor, barrier-bracketed K steps, max over ranks, ranks_seen -- with
    the reference's CPU stand-in executor (src/infer/random.cc via nsg_cpu_executor) in the evaluator's place."""
    import torch.distributed as dist
    nsg = importlib.import_module("nshogi-engine_amd")
    distributed = world > 1
    if distributed:
        dist.init_process_group(backend="gloo")
    blocks, channels = (int(x) for x in args.net.split("x"))
    blob = nsg.weights.to_blob(nsg.weights.make_random(1, 64, seed=0)) if rank == 0 else None
    blob_bytes = len(blob) if blob is not None else 0
    if distributed:
        got = nsg.dist.broadcast_blob(blob, src=0, device="cpu")
        blob_bytes = int(got.numel())
    ranks_seen = int(nsg.dist.sum_over_ranks(1.0)) if distributed else 1
    B = args.batch
    ex = nsg.CpuExecutor("random", seed=nsg.dist.shard_seed(0, rank))
    pol = np.empty((B, 2187), np.float32)
    win = np.empty(B, np.float32)
    drw = np.empty(B, np.float32)

    def barrier():
        if distributed:
            dist.barrier()

    for _ in range(args.warmup):
        ex.compute_blocking(B, policy=pol, win=win, draw=drw)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ex.compute_blocking(B, policy=pol, win=win, draw=drw)
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        dt = nsg.dist.max_over_ranks(dt)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": f"NN evals/sec at batch={B}", "value": B * args.steps * world / dt, "unit": "evals/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "executor": "random",
            "config": {"workload": f"CPU REHEARSAL of the multi-rank path, not the benchmark: batch={B}, "
                                   f"infer::Random stand-in executor on the host, {world} rank(s) over gloo",
                       "batch_per_gpu": B, "weights_broadcast_bytes": blob_bytes}}))
    return 0


def sample_clock(gpu, out, stop, period=1.0):
    """rocm-smi sclk / socket power of `gpu` every `period` s until `stop` is set (a child process per sample)."""
    import re
    import subprocess
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", str(gpu), "--showclocks", "--showpower"],
                               capture_output=True, text=True, timeout=20)
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", r.stdout)
            w = re.search(r"Power \(W\): ([0-9.]+)", r.stdout)
            if m:
                out.append((float(m.group(1)), float(w.group(1)) if w else None))
        except (OSError, subprocess.SubprocessError):
            return
        stop.wait(period)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--net", default="20x256", help="blocks x channels, e.g. 10x192, 20x256, 40x384")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--precision", default=os.environ.get("NSG_BENCH_PRECISION", "f16m6"),
                    choices=["fp32", "fp16", "bf16", "f16x3", "f16m8", "f16m6"])
    ap.add_argument("--positions", default="startpos", choices=["startpos", "distinct", "synthetic"],
                    help="startpos: the initial position in every slot (bench/batchsize.cc:47-59, the default); "
                         "distinct: B distinct positions of random-playout games; synthetic: seeded random bitboards")
    ap.add_argument("--sustain-seconds", type=float, default=5.0,
                    help="length of the sustained (>= 5 s) legs; 0 disables them")
    ap.add_argument("--selfplay-seconds", type=float, default=60.0,
                    help="length of the self-play leg (metric #2, games/sec); 0 disables it")
    # BASELINE configs[3]: 256 concurrent games per GPU = 1 thread x 2 groups x 128 games
    # (profiles/r01/h_selfplay_shape_256_games.txt)
    ap.add_argument("--selfplay-threads", type=int, default=1)
    # host threads advancing the engine's games between two batches: one keeps up in the opening, six to eight are
    # needed once positions get busy -- a leaf then costs 10-30 us of move generation and mate search
    # (profiles/r02/a_selfplay_shape_workers.txt, e_selfplay_60s_workers_solvers.txt)
    ap.add_argument("--selfplay-workers", type=int, default=8)
    # threads that run the df-pn mate solver of judge (100 000 nodes, worker.cc:516) off the search path
    ap.add_argument("--selfplay-solver-threads", type=int, default=4)
    ap.add_argument("--selfplay-games-per-group", type=int, default=128)
    ap.add_argument("--selfplay-deterministic-seconds", type=float, default=20.0,
                    help="length of the second self-play leg in the shape whose move digests reproduce bit for bit "
                         "(one search worker, no solver pool: north_star's fixed-seed determinism); 0 disables it")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the by_config legs (configs[1], configs[4])")
    ap.add_argument("--workload-only", type=int, default=0, metavar="N",
                    help="profiler workload: load, upload, N device-resident steps of the benchmark's batch, one read-back, "
                         "exit -- no HIP events, no child processes, no other legs (scripts/pmc.sh runs rocprofv3 --pmc "
                         "passes over this; add --hip-events to bracket the launches with events as the timed run does)")
    ap.add_argument("--hip-events", action="store_true", help="with --workload-only: enable the HIP-event timing too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-sample", action="store_true",
                    help="no rocm-smi child processes during the sustained leg (under rocprofv3 the box refuses a "
                         "profiled process's children that exec another program)")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--executor", default="hip", choices=["hip", "random"],
                    help="hip: the MI355X evaluator (the benchmark).  random: CPU rehearsal of the multi-rank path -- "
                         "gloo backend, the reference's EXECUTOR=random stand-in as the executor, no GPU touched")
    args = ap.parse_args()

    # ---- ranks.  The driver starts N ranks with torch.distributed.run; started WITHOUT a launcher and with
    # --gpus N > 1 this process becomes the launcher (the reference multiplies executors from one command,
    # selfplay/main.cc:33,189-195): it touches neither the GPU nor libnsg.so, starts the ranks as a child
    # process and relays their line.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                  f"run `python bench.py --gpus N` (it starts N ranks itself) or give torch.distributed.run "
                  f"--nproc-per-node equal to --gpus", file=sys.stderr)
        sys.exit(2)
    if args.executor == "random":
        sys.exit(rehearse_cpu(args, rank, world))

    import torch
    import torch.distributed as dist

    # (NSG_BENCH_FORCE_DIST=1: take the multi-rank code path -- RCCL weight broadcast, reductions -- with one rank,
    # to rehearse it on a one-GPU box)
    distributed = world > 1 or os.environ.get("NSG_BENCH_FORCE_DIST") == "1"
    if distributed:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    nsg = importlib.import_module("nshogi-engine_amd")
    blocks, channels = (int(x) for x in args.net.split("x"))
    B = args.batch
    ranks_seen = int(round(nsg.dist.sum_over_ranks(1.0, device="cuda"))) if distributed else 1

    ev = nsg.Evaluator(local_rank, B, 86, precision=args.precision)

    # weights: rank 0 builds the synthetic blob; one RCCL broadcast over xGMI
    # replaces every executor re-reading the model file (trt.cc:109-186).
    blob = None
    if rank == 0:
        blob = nsg.weights.to_blob(nsg.weights.make_random(blocks, channels, seed=0, bn="identity"))
    if distributed:
        dev_blob = nsg.dist.broadcast_blob(blob, src=0, device="cuda")
        torch.cuda.synchronize()
        ev.load_device_blob(dev_blob.data_ptr(), dev_blob.numel())
        if blob is None:  # the self-play leg hands this rank's driver the same bytes
            blob = dev_blob.cpu().numpy().tobytes()
        del dev_blob
    else:
        ev.load_memory(blob)
    info = ev.info()

    # positions (per rank): the reference benchmark's input, or distinct real / synthetic ones
    def positions(kind):
        if kind == "startpos":
            return nsg.positions.startpos_batch(B)
        if kind == "distinct":
            return nsg.positions.game_positions(B, seed=nsg.dist.shard_seed(nsg.synth.SEED, rank))
        return nsg.synth.random_batch(B, 86, seed=nsg.dist.shard_seed(nsg.synth.SEED, rank), distinct=True)

    bb = positions(args.positions)
    ev.upload_features(bb)
    if args.workload_only > 0:
        if args.hip_events:
            ev.profile_enable(True)
        for _ in range(args.workload_only):
            ev.forward_resident(B)
        p, v, d = ev.download_outputs(B)
        if args.hip_events:
            print(json.dumps(ev.profile_read()))
        print(json.dumps({"workload_only": args.workload_only, "batch": B, "net": args.net, "precision": args.precision,
                          "policy_max": float(p.max()), "value0": float(v[0]), "plan": ev.last_plan()}))
        ev.close()
        if distributed:
            dist.destroy_process_group()
        return

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # The sustained leg runs FIRST: the same loop held for >= 5 s (the kernel is power-limited and the chip
    # needs a few hundred milliseconds of load to reach its steady clocks), HIP-event timing of the trunk
    # conv over that whole window.  The W warm-up and K timed steps of the measurement contract follow on a
    # chip that is already at those clocks, as it is in an engine that evaluates continuously; from a cold
    # chip the K = 20 steps (60 ms) read 2-3 % below the sustained rate.
    ev.profile_enable(True)
    sustained = None
    if args.sustain_seconds > 0:
        for _ in range(max(args.warmup, 1)):
            ev.forward_resident(B)
        barrier()
        ev.profile_read()
        import threading
        clock_samples, clock_stop = [], threading.Event()
        sampler = None
        profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        if rank == 0 and not args.no_clock_sample and not profiled:
            # the chip runs this kernel power-limited: the clock it HOLDS under the load is sampled live
            sampler = threading.Thread(target=sample_clock, args=(local_rank, clock_samples, clock_stop), daemon=True)
            sampler.start()
        rate, n, secs = held_rate(lambda: ev.forward_resident(B), B, torch.cuda.synchronize, args.sustain_seconds)
        clock_stop.set()
        if sampler is not None:
            sampler.join(timeout=30)
        sprof = ev.profile_read()
        if distributed:
            rate = nsg.dist.sum_over_ranks(rate, device="cuda")
        sustained = {"evals_per_sec": rate, "steps": n, "seconds": secs,
                     "conv_avg_launch_ms": sprof["trunk_ms_total"] / max(sprof["trunk_launches"], 1),
                     "conv_launches_timed": sprof["trunk_launches"]}
        mhz = [c for c, _ in clock_samples[1:]] or [c for c, _ in clock_samples]  # the first sample may predate the load
        if mhz:
            watts = [w for _, w in clock_samples if w is not None]
            sustained["sclk_mhz_observed"] = sum(mhz) / len(mhz)
            sustained["sclk_samples"] = len(mhz)
            sustained["socket_power_w_observed"] = sum(watts) / len(watts) if watts else None

    for _ in range(args.warmup):
        ev.forward_resident(B)
    barrier()
    ev.profile_read()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.forward_resident(B)
    barrier()
    dt = time.perf_counter() - t0
    prof = ev.profile_read()

    if distributed:
        dt = nsg.dist.max_over_ranks(dt, device="cuda")
    ev.profile_enable(False)

    # ---- metric #2: self-play games/sec (each rank drives its own GPU; games shard
    # embarrassingly, no collective on the data path)
    sp = None
    if args.selfplay_seconds > 0 and os.path.exists(SELFPLAY_BIN):
        import tempfile
        fd, wpath = tempfile.mkstemp(prefix=f"nsg_bench_weights_r{rank}_", suffix=".nsgw")
        with os.fdopen(fd, "wb") as f:  # every rank writes its own copy of the broadcast bytes
            f.write(blob)
        barrier()
        ev.close()  # free this process's evaluator before the self-play process allocates its own
        mine = selfplay_leg(wpath, local_rank, args.selfplay_seconds, args.selfplay_threads, args.precision,
                            games_per_group=args.selfplay_games_per_group, workers=args.selfplay_workers,
                            solvers=args.selfplay_solver_threads)
        det = None
        if args.selfplay_deterministic_seconds > 0 and world == 1:
            # the same leg in the shape whose per-game move sequences are a function of the seed alone
            # (selfplay.h: one search worker, no solver pool): what that guarantee costs in throughput
            det = selfplay_leg(wpath, local_rank, args.selfplay_deterministic_seconds, args.selfplay_threads,
                               args.precision, games_per_group=args.selfplay_games_per_group, workers=1, solvers=0)
        try:
            os.remove(wpath)
        except OSError:
            pass
        keys = ("games_per_sec", "games_per_sec_window", "moves_per_sec", "playouts_per_sec", "evals_per_sec",
                "games_finished", "concurrent_games")
        # every rank takes part in every collective, whether its self-play process failed or not
        failed = 1.0 if "error" in mine else 0.0
        if distributed:
            failed = nsg.dist.sum_over_ranks(failed, device="cuda")
            tot = {k: nsg.dist.sum_over_ranks(float(mine.get(k, 0.0)), device="cuda") for k in keys}
        else:
            tot = {k: mine.get(k, 0.0) for k in keys}
        if failed:
            sp = {"error": mine.get("error", "the self-play process failed on another rank"), "ranks_failed": int(failed)}
        else:
            sp = dict(tot, **{k: mine[k] for k in ("avg_batch", "cache_hit_ratio", "avg_game_length", "playouts_per_move",
                                                   "seconds", "window_seconds", "await_ms_per_batch", "host_ms_per_batch",
                                                   "batches_found_finished") if k in mine},
                      threads_per_gpu=mine.get("threads"), workers_per_thread=mine.get("workers"),
                      solver_threads=mine.get("solver_threads"),
                      note="AlphaZero-mode self-play from startpos on this build's own shogi core; synthetic "
                           "(untrained) weights, so games end early by repetition: games/sec is a plumbing number, "
                           "evals/playouts per sec are the load.  games_per_sec = finished / elapsed from a cold "
                           "start (saveworker.cc:135-137); games_per_sec_window = games finished in the second "
                           "half of the run / its length (the cold start excluded).  await_ms_per_batch = engine thread "
                           "blocked in Infer::await, host_ms_per_batch = its work between two batches, "
                           "batches_found_finished = share of batches that had already finished when the engine came "
                           "back for them (0: the executor never waited for the host)")
            sp["determinism"] = ("this shape (%s search workers, %s solver threads) is the throughput shape: batch "
                                 "composition depends on thread timing, so per-game move digests do NOT reproduce from "
                                 "the seed; the deterministic_shape leg (1 worker, no solver pool) is the one that "
                                 "satisfies north_star's 'bit-identical under a fixed RNG seed' "
                                 "(tests: test_selfplay_hip_reproducible, test_config3_selfplay_256_games_800_playouts)"
                                 % (mine.get("workers"), mine.get("solver_threads")))
            if det is not None and "error" not in det:
                sp["deterministic_shape"] = {k: det[k] for k in ("games_per_sec", "games_per_sec_window", "evals_per_sec",
                                                                 "playouts_per_sec", "avg_batch", "seconds", "digest",
                                                                 "games_finished", "await_ms_per_batch", "host_ms_per_batch")
                                             if k in det}
                sp["deterministic_shape"].update(workers_per_thread=1, solver_threads=0)
            elif det is not None:
                sp["deterministic_shape"] = det
        barrier()
        ev = None

    # ---- the reference's own metric (bench/batchsize.cc:61-79) on EVERY rank at once: 4 warm-ups, then back-to-back
    # computeBlocking incl. H2D / D2H on the initial position replicated B times; whole job = sum over ranks
    ref = None
    hostev = None
    if not args.no_host_path:
        hostev = nsg.Evaluator(local_rank, B, 86, precision=args.precision)
        hostev.load_memory(blob)
        pol = np.empty((B, 2187), np.float32)
        win = np.empty(B, np.float32)
        drw = np.empty(B, np.float32)
        secs = max(args.sustain_seconds, 1.0)
        start = bb if args.positions == "startpos" else nsg.positions.startpos_batch(B)
        for _ in range(4):
            hostev.compute_blocking(start, policy=pol, win=win, draw=drw)
        barrier()
        rate, n, took = held_rate(lambda: hostev.compute_blocking(start, policy=pol, win=win, draw=drw), B,
                                  lambda: None, secs)
        if distributed:
            rate = nsg.dist.sum_over_ranks(rate, device="cuda")
        ref = {"definition": "bench/batchsize.cc:61-79: BatchSize * Repeat / wall over back-to-back "
                             "computeBlocking (H2D + planes + net + D2H + sync), 4 warm-ups, the initial "
                             "position in every slot" + ("; sum over ranks, all ranks running at once" if world > 1 else ""),
               "evals_per_sec": rate, "repeat": n, "seconds": took}
        if world > 1:
            hostev.close()
            hostev = None

    out = None
    if rank == 0:
        evals = B * args.steps * world
        value = evals / dt
        flops_pos = info["flops_per_position"]
        # The timed launches are the 2N F->F convolutions of a forward -- or, when the evaluator ran the whole
        # trunk as ONE persistent launch (f16m6 at this batch: nsg.h, nsg_profile_read), that launch: stem + 2N convs.
        persistent = prof["trunk_launches"] == prof["forwards"]
        stem_flops = 2.0 * 81 * 9 * 86 * channels
        conv_flops_launch = info["trunk_conv_flops_per_position"] * B
        if persistent:
            conv_flops_launch = (stem_flops + 2 * blocks * info["trunk_conv_flops_per_position"]) * B
        conv_ms = prof["trunk_ms_total"] / max(prof["trunk_launches"], 1)
        achieved = conv_flops_launch / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        pmc, pmc_file = pmc_summary(args, B)
        what = {"startpos": "the initial position in every slot (bench/batchsize.cc:47-59)",
                "distinct": "B distinct positions of random-playout games", "synthetic": "seeded random bitboards"}
        out = {
            "metric": "NN evals/sec at batch=512" if B == 512 else f"NN evals/sec at batch={B}",
            "value": value, "unit": "evals/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "value_is": "device-resident rate (bitboards in HBM, outputs left in HBM); the reference's own "
                        "PCIe-inclusive definition is reference_metric.evals_per_sec; the timed steps follow the "
                        "sustained leg, i.e. run at the chip's steady clocks",
            "config": {"workload": f"batch={B} x {blocks}-block x {channels}-channel policy/value/draw "
                                   f"resnet, 86 feature planes, device-resident bitboards -> planes -> "
                                   f"trunk -> heads (BASELINE configs[2] evaluator leg); input: {what[args.positions]}",
                       "batch_per_gpu": B, "net": args.net, "precision": args.precision, "positions": args.positions,
                       "parallelism": f"{world} independent evaluators (positions sharded, no data-path collective)",
                       "weights": "synthetic He-normal seed 0, BN folded, broadcast from rank 0"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": conv_traffic_bytes(pmc),
                         "traffic_source": (f"{pmc_file} (committed rocprofv3 --pmc passes of this workload, NOT "
                                            f"measured by this run)" if pmc_file else None),
                         "mfma_pipe_busy_frac_pmc": conv_mfma_busy(pmc),
                         "mfma_flops_executed_per_algorithmic_flop": MFMA_UNITS[args.precision],
                         "kernel": ("trunkKernel (ONE persistent launch: stem + %d x 3x3 conv F->F, bias+residual+ReLU fused)" % (2 * blocks))
                                   if persistent else "tileKernel<kConv> (3x3 conv F->F, bias+residual+ReLU fused)",
                         "avg_launch_ms": conv_ms, "launches_timed": prof["trunk_launches"],
                         "algorithmic_flops_per_launch": conv_flops_launch},
            "whole_net_tflops": value / world * flops_pos / 1e12,
            "whole_net_frac_of_peak": value / world * flops_pos / 1e12 / peak,
            "forward_ms_hip_events": prof["forward_ms_total"] / max(prof["forwards"], 1),
            "device": info["device_name"], "compute_units": info["compute_units"],
        }
        if sustained is not None and sustained.get("sclk_mhz_observed"):
            # the 2516.6 TF peak assumes the 2.4 GHz boost clock; under a 16-bit MFMA load the chip is power-limited
            # and holds less -- the same fraction against the clock it actually held
            pk = peak * sustained["sclk_mhz_observed"] / 2400.0
            out["roofline"]["peak_at_observed_clock"] = pk
            out["roofline"]["frac_at_observed_clock"] = achieved / pk
            out["roofline"]["observed_clock_mhz"] = sustained["sclk_mhz_observed"]
        if sustained is not None:
            out["sustained_evals_per_sec"] = sustained["evals_per_sec"]
            sc = sustained["conv_avg_launch_ms"]
            sustained["conv_frac_of_peak"] = conv_flops_launch / (sc * 1e-3) / 1e12 / peak if sc > 0 else None
            sustained["persistent_trunk_launch"] = bool(persistent)
            out["sustained"] = sustained
        if sp is not None:
            out["selfplay"] = sp
        if ref is not None:
            ref["fraction_of_device_resident"] = ref["evals_per_sec"] / (sustained["evals_per_sec"] if sustained else value)
            out["reference_metric"] = ref
            out["host_path_evals_per_sec"] = ref["evals_per_sec"]
        if hostev is not None and world == 1:
            ev = hostev
            # the B-distinct variant of both rates (real positions of random-playout games)
            other = positions("distinct" if args.positions != "distinct" else "startpos")
            label = "distinct_positions" if args.positions != "distinct" else "startpos"
            rate, n, took = held_rate(lambda: ev.compute_blocking(other, policy=pol, win=win, draw=drw), B,
                                      lambda: None, secs)
            ref[f"evals_per_sec_{label}"] = rate
            ev.upload_features(other)
            rate, n, took = held_rate(lambda: ev.forward_resident(B), B, torch.cuda.synchronize, secs)
            out[f"device_resident_evals_per_sec_{label}"] = rate
            # the same with the legal-move lookup on the device: 80 legal moves per
            # position, softmax priors returned -- D2H shrinks from 8748 B to 320 B per position
            rng = np.random.default_rng(1)
            off = (np.arange(B + 1) * 80).astype(np.uint32)
            idx = rng.integers(0, 2187, size=B * 80).astype(np.uint16)
            vals = np.empty(B * 80, np.float32)
            rate, n, took = held_rate(lambda: ev.compute_gather_blocking(start, idx, off, softmax=True, values=vals,
                                                                         win=win, draw=drw), B, lambda: None, min(secs, 2.0))
            out["host_path_device_gather_evals_per_sec"] = rate
            ev.close()
            # the evaluator at the other batch sizes the survey asks for (device-resident)
            if B == 512 and args.net == "20x256":
                big = nsg.Evaluator(local_rank, 1024, 86, precision=args.precision)
                big.load_memory(blob)
                big.upload_features(nsg.positions.startpos_batch(1024))
                by_batch, prec_by_batch = {}, {}
                for nb in (1, 8, 16, 17, 32, 64, 128, 256, 1024):  # 128 = the engine's default BatchSize (context.h:79)
                    rate, n, took = held_rate(lambda: big.forward_resident(nb), nb, torch.cuda.synchronize, 1.0)
                    by_batch[str(nb)] = rate
                    prec_by_batch[str(nb)] = big.last_plan()["trunk_precision"]
                by_batch["512"] = value
                prec_by_batch["512"] = args.precision
                out["evals_per_sec_by_batch"] = by_batch
                # the arithmetic each batch size ran its trunk in: up to sixteen boards take the team trunk (f16x3: three
                # f16 MFMAs per MAC on 4-byte-per-weight hi/lo records), the rest the line's precision
                out["trunk_precision_by_batch"] = prec_by_batch
                # ... and as a fraction of the MFMA roofline (north_star: "as absolute numbers and as fraction
                # of the MFMA roofline"): evals/s x algorithmic flops per position / the f16 dense peak
                out["frac_of_mfma_roofline_by_batch"] = {k: v * flops_pos / 1e12 / peak for k, v in by_batch.items()}
                # small batches are bound by the WEIGHT stream, not by MFMA: every forward reads every packed trunk
                # record once (4 bytes per weight as packed -- MX plans: f16 + two e2m3 copies + exponents / padding; the
                # team trunk's f16x3 records: f16 hi + f16 lo -- the same 4 bytes in either format), so the roof
                # is forwards/s x packed bytes against the 8 TB/s of HBM (the weights sit in the 256 MB Infinity Cache
                # between forwards; HBM is the conservative roof)
                wbytes = 4.0 * 9 * (128 * channels + 2 * blocks * channels * channels)
                out["packed_trunk_weight_bytes_per_forward"] = wbytes
                out["frac_of_weight_bw_roofline_by_batch"] = {k: (v / int(k)) * wbytes / 8e12 for k, v in by_batch.items()
                                                              if int(k) <= 64}
                big.close()
            # ---- the other BASELINE configs on this GPU with this round's kernels
            if B == 512 and args.net == "20x256" and not args.no_other_configs:
                out["by_config"] = {
                    "configs[1] 10x192 batch 64": config_leg(nsg, local_rank, "10x192", 64, args.precision),
                    "configs[4] 40x384 bf16 batch 1024": config_leg(nsg, local_rank, "40x384", 1024, "bf16"),
                    "configs[4] 40x384 batch 1024 at the 1e-3 parity arithmetic": config_leg(nsg, local_rank, "40x384", 1024,
                                                                                       args.precision)}
            rev = nsg.Evaluator(local_rank, B, 86, precision=args.precision)
            rev.load_memory(blob)
            rev.upload_features(bb)
            out["roofline_extract"] = extract_roofline(nsg, rev, bb, B, args.precision)
            rev.close()
        if not args.no_host_path and world == 1:
            out["other_precisions_evals_per_sec"] = {
                p: quick_rate(nsg, local_rank, blob, bb, B, p) for p in ("fp32", "f16x3", "f16m8", "f16m6", "fp16", "bf16")
                if p != args.precision}
        if not args.no_cpu_baseline and world == 1:
            # N = 1 only (the measurement contract): at N > 1 the other ranks would spin in a collective on the host
            # cores this leg measures
            out["cpu_baseline"] = cpu_baseline(nsg)
            if os.path.exists(SELFPLAY_BIN):
                out["cpu_baseline"]["selfplay_random_1thread_100playouts"] = selfplay_cpu_baseline()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
