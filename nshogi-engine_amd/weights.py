"""NSGW v1 weight files (DESIGN.md "Weight file") and synthetic weights.

The reference ships no model: it loads an arbitrary ONNX file through
TensorRT (src/infer/trt.cc:109-232) and only fixes the tensor contract
(input [N,C,9,9]; outputs policy[2187], value, draw).  The topology, this
file format and the synthetic initialisation are therefore the build's own.
"""
import struct

import numpy as np

MAGIC = b"NSGW"
VERSION = 1
HEADER_BYTES = 64
BN_EPS = 1e-5

# Named nets of BASELINE.json's configs: (blocks, channels)
NETS = {"10x192": (10, 192), "20x256": (20, 256), "40x384": (40, 384)}


def tensor_order(blocks):
    names = ["stem_w", "stem_bn"]
    for k in range(blocks):
        names += [f"b{k}_w1", f"b{k}_bn1", f"b{k}_w2", f"b{k}_bn2"]
    names += ["policy_w", "policy_b", "value_w", "value_bn", "fc1_w", "fc1_b", "fc2_w", "fc2_b"]
    return names


def shapes(blocks, channels, in_channels=86, policy_channels=27, value_channels=32,
           value_hidden=256):
    F, C, VC, VH, PC = channels, in_channels, value_channels, value_hidden, policy_channels
    s = {"stem_w": (F, C, 3, 3), "stem_bn": (4, F)}
    for k in range(blocks):
        s[f"b{k}_w1"] = (F, F, 3, 3)
        s[f"b{k}_bn1"] = (4, F)
        s[f"b{k}_w2"] = (F, F, 3, 3)
        s[f"b{k}_bn2"] = (4, F)
    s.update({"policy_w": (PC, F), "policy_b": (PC,), "value_w": (VC, F), "value_bn": (4, VC),
              "fc1_w": (VH, VC * 81), "fc1_b": (VH,), "fc2_w": (2, VH), "fc2_b": (2,)})
    return s


def flops_per_position(blocks, channels, in_channels=86):
    """stem + residual trunk + 1x1 policy (value/draw heads excluded)."""
    F = channels
    return 2 * 81 * 9 * in_channels * F + blocks * 2 * (2 * 81 * 9 * F * F) + 2 * 81 * 27 * F


def make_random(blocks, channels, in_channels=86, value_channels=32, value_hidden=256, seed=0,
                bn="identity"):
    """He-normal convolutions.  bn="identity": gamma=1, beta=0, mean=0, var=1
    (SURVEY.md 8d), except the second BN of every residual block whose gamma
    is 1/sqrt(blocks) so activations stay O(1) through deep trunks;
    bn="random": every BN statistic randomised (exercises the folding)."""
    rng = np.random.default_rng(seed)
    sh = shapes(blocks, channels, in_channels, 27, value_channels, value_hidden)
    w = {}

    def he(shape, fan_in):
        return (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)

    def bnp(n, gamma=1.0):
        if bn == "identity":
            return np.stack([np.full(n, gamma), np.zeros(n), np.zeros(n), np.ones(n)]).astype(np.float32)
        return np.stack([gamma * rng.uniform(0.5, 1.5, n), rng.normal(0, 0.1, n),
                         rng.normal(0, 0.1, n), rng.uniform(0.5, 1.5, n)]).astype(np.float32)

    w["stem_w"] = he(sh["stem_w"], in_channels * 9)
    w["stem_bn"] = bnp(channels)
    g2 = 1.0 / np.sqrt(max(blocks, 1))
    for k in range(blocks):
        w[f"b{k}_w1"] = he(sh[f"b{k}_w1"], channels * 9)
        w[f"b{k}_bn1"] = bnp(channels)
        w[f"b{k}_w2"] = he(sh[f"b{k}_w2"], channels * 9)
        w[f"b{k}_bn2"] = bnp(channels, g2)
    w["policy_w"] = he(sh["policy_w"], channels)
    w["policy_b"] = rng.normal(0, 0.1, 27).astype(np.float32)
    w["value_w"] = he(sh["value_w"], channels)
    w["value_bn"] = bnp(value_channels)
    w["fc1_w"] = he(sh["fc1_w"], value_channels * 81)
    w["fc1_b"] = rng.normal(0, 0.1, value_hidden).astype(np.float32)
    w["fc2_w"] = (rng.standard_normal((2, value_hidden)) * np.sqrt(1.0 / value_hidden)).astype(np.float32)
    w["fc2_b"] = rng.normal(0, 0.1, 2).astype(np.float32)
    w["_meta"] = dict(blocks=blocks, channels=channels, in_channels=in_channels,
                      policy_channels=27, value_channels=value_channels,
                      value_hidden=value_hidden, bn_eps=BN_EPS)
    return w


def to_blob(w):
    m = w["_meta"]
    hdr = MAGIC + struct.pack("<7If", VERSION, m["in_channels"], m["channels"], m["blocks"],
                              m["policy_channels"], m["value_channels"], m["value_hidden"],
                              m["bn_eps"])
    hdr += b"\0" * (HEADER_BYTES - len(hdr))
    sh = shapes(m["blocks"], m["channels"], m["in_channels"], m["policy_channels"],
                m["value_channels"], m["value_hidden"])
    parts = [hdr]
    for name in tensor_order(m["blocks"]):
        a = np.ascontiguousarray(w[name], dtype="<f4")
        if a.shape != sh[name]:
            raise ValueError(f"{name}: shape {a.shape}, expected {sh[name]}")
        parts.append(a.tobytes())
    return b"".join(parts)


def from_blob(blob):
    if blob[:4] != MAGIC:
        raise ValueError("not an NSGW file")
    version, cin, F, blocks, pc, vc, vh, eps = struct.unpack("<7If", blob[4:36])
    if version != VERSION:
        raise ValueError(f"unsupported NSGW version {version}")
    sh = shapes(blocks, F, cin, pc, vc, vh)
    w = {"_meta": dict(blocks=blocks, channels=F, in_channels=cin, policy_channels=pc,
                       value_channels=vc, value_hidden=vh, bn_eps=eps)}
    off = HEADER_BYTES
    for name in tensor_order(blocks):
        n = int(np.prod(sh[name]))
        w[name] = np.frombuffer(blob, dtype="<f4", count=n, offset=off).reshape(sh[name]).copy()
        off += 4 * n
    if off != len(blob):
        raise ValueError("NSGW size mismatch")
    return w


def save(path, w):
    with open(path, "wb") as f:
        f.write(to_blob(w))


def load(path):
    with open(path, "rb") as f:
        return from_blob(f.read())
