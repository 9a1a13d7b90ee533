#!/usr/bin/env python3
"""What `Infer::load` costs (the decision behind UseSerializedFileIfAvailable, /root/reference/src/infer/trt.cc:111-119,
171-186: TensorRT caches its built engine because building one takes minutes).  Times, per net and arithmetic:
  * nsg_convert_onnx: ONNX bytes -> NSGW (protobuf decode, topology check, BN kept unfolded);
  * nsg_load_memory on the ONNX bytes: convert + BN folding in double + fragment packing (worker threads) + upload;
  * nsg_load_shared: a second evaluator of the same device adopting the packed weights.
scripts/load_time.py [--nets 20x256,40x384] [--precisions f16m6,bf16,fp32]"""
import argparse, importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nsg = importlib.import_module("nshogi-engine_amd")
ap = argparse.ArgumentParser(); ap.add_argument("--nets", default="20x256,40x384"); ap.add_argument("--precisions", default="f16m6,bf16,fp32")
a = ap.parse_args()
out = []
for net in a.nets.split(","):
    blocks, ch = (int(x) for x in net.split("x"))
    w = nsg.weights.make_random(blocks, ch, seed=0)
    onnx = nsg.onnx_io.export_onnx(w)
    t0 = time.perf_counter(); blob = nsg.convert_onnx(onnx); t_conv = time.perf_counter() - t0
    for prec in a.precisions.split(","):
        ev = nsg.Evaluator(0, 512, 86, precision=prec)
        t0 = time.perf_counter(); ev.load_memory(onnx); t_load = time.perf_counter() - t0
        ev2 = nsg.Evaluator(0, 512, 86, precision=prec)
        t0 = time.perf_counter(); ev2.load_shared(ev); t_shared = time.perf_counter() - t0
        row = {"net": net, "precision": prec, "onnx_MB": round(len(onnx) / 1e6, 1), "convert_onnx_s": round(t_conv, 3),
               "load_from_onnx_s": round(t_load, 3), "load_shared_s": round(t_shared, 4), "host_threads": os.cpu_count()}
        print(json.dumps(row), flush=True)
        out.append(row)
        ev2.close(); ev.close()
