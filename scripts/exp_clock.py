#!/usr/bin/env python3
"""Same-box comparison of libnsg builds with the clock the chip HOLDS under each:
scripts/exp_clock.py [--batch 512] [--seconds 4] [--rounds 2] name=path.so ...
Each (round, library) is its own process: a sustained device-resident loop while the parent samples
rocm-smi (sclk, socket power).  Prints evals/s, conv ms by HIP events, MHz, W per run."""
import argparse, json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, os, sys, time
sys.path.insert(0, %(root)r)
import torch
nsg = importlib.import_module("nshogi-engine_amd")
B = %(batch)d
ev = nsg.Evaluator(0, B, 86, precision=%(prec)r)
ev.load_memory(nsg.weights.to_blob(nsg.weights.make_random(20, 256, seed=0, bn="identity")))
ev.upload_features(nsg.positions.game_positions(B, seed=9))
for _ in range(5): ev.forward_resident(B)
torch.cuda.synchronize()
if %(flags)d: os.environ["NSG_EXP_FLAGS"] = str(%(flags)d)  # from here on: the timing-only variant, on the buffers the correct forwards left
ev.profile_enable(True); ev.profile_read()
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < %(seconds)f:
    for _ in range(8): ev.forward_resident(B)
    n += 8
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p = ev.profile_read()
print(json.dumps({"evals_per_sec": B * n / dt, "conv_ms": p["trunk_ms_total"] / max(p["trunk_launches"], 1)}))
'''
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+"); ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--seconds", type=float, default=4.0); ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--precision", default="f16m6")
a = ap.parse_args()
libs = [x.split("=", 1) for x in a.libs]  # name=path[:flags]
libs = [(n, p.split(":")[0], int(p.split(":")[1]) if ":" in p else 0) for n, p in libs]


def sample(out, stop):
    while not stop.is_set():
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", r.stdout); w = re.search(r"Power \(W\): ([0-9.]+)", r.stdout)
        if m: out.append((float(m.group(1)), float(w.group(1)) if w else 0.0))
        stop.wait(0.5)


acc = {n: [] for n, _, _ in libs}
for rnd in range(a.rounds):
    for name, path, flags in libs:
        env = dict(os.environ, NSG_LIB=os.path.abspath(path))
        child = subprocess.Popen([sys.executable, "-c", CHILD % {"root": ROOT, "batch": a.batch, "prec": a.precision, "seconds": a.seconds, "flags": flags}],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        time.sleep(max(3.0, 0.0))  # let the child load and warm up (import torch, weights)
        out, stop = [], threading.Event()
        th = threading.Thread(target=sample, args=(out, stop)); th.start()
        so, se = child.communicate(timeout=600)
        stop.set(); th.join()
        try:
            d = json.loads(so.strip().split("\n")[-1])
        except Exception:
            print(name, "FAILED", se[-400:]); continue
        busy = [x for x in out if x[1] > 600] or out  # samples taken under load
        d["mhz"] = sum(x[0] for x in busy) / max(len(busy), 1); d["watts"] = sum(x[1] for x in busy) / max(len(busy), 1); d["samples"] = len(busy)
        acc[name].append(d)
        print(f"round {rnd} {name:24s} {d['evals_per_sec']:10.0f} evals/s  conv {d['conv_ms']*1e3:7.2f} us  {d['mhz']:6.0f} MHz {d['watts']:6.0f} W ({d['samples']} samples)", flush=True)
print(json.dumps(acc))
