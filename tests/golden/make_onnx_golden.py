"""Generates tests/golden/net_torch_2x64.onnx, net_torch_bn_1x64.onnx, net_torch_sigtanh_eps_1x64.onnx and net_torch.npz (run from the repo root:
`python tests/golden/make_onnx_golden.py`).

The engine hands its executor an ONNX file (src/infer/trt.cc:109-131, default ./res/model.onnx,
src/context.h:93).  The reference ships no model, so this script makes one with a REAL producer:
a torch.nn.Module of the topology of DESIGN.md section 2 (random weights and BatchNorm statistics,
seed 20240203), serialised by PyTorch's own ONNX exporter (torch.onnx.export, TorchScript
exporter, opset 17).  The npz holds 6 real positions (feature bitboards from the build's rules
core), their expanded planes, and the module's float64 forward outputs -- the expectation the
ONNX readers (nsg_convert_onnx in libnsg.so, onnx_io.import_onnx) are held to.  Two files: the
usual eval-mode export (BatchNorm folded into conv weight + bias by the exporter) and one that
keeps the BatchNormalization nodes.

Note on the exporter: torch 2.10's legacy exporter calls a post-step that imports the `onnx`
package only to splice onnxscript custom functions into the proto; there are none here and the
package is not in this image, so that step is replaced by the identity.  The ModelProto bytes
come from torch's C++ serialiser (graph._export_onnx) untouched.
"""
import io
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BLOCKS, F, C, VC, VH = 2, 64, 86, 8, 64


class Block(nn.Module):
    def __init__(self, f):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(f)
        self.conv2 = nn.Conv2d(f, f, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(f)

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return torch.relu(x + y)


class Net(nn.Module):
    def __init__(self, BLOCKS=BLOCKS, swapped_squashing=False):
        super().__init__()
        self.swapped_squashing = swapped_squashing
        self.stem = nn.Conv2d(C, F, 3, padding=1, bias=False)
        self.stem_bn = nn.BatchNorm2d(F)
        self.blocks = nn.ModuleList([Block(F) for _ in range(BLOCKS)])
        self.policy = nn.Conv2d(F, 27, 1)
        self.value_conv = nn.Conv2d(F, VC, 1, bias=False)
        self.value_bn = nn.BatchNorm2d(VC)
        self.fc1 = nn.Linear(VC * 81, VH)
        self.fc_value = nn.Linear(VH, 1)
        self.fc_draw = nn.Linear(VH, 1)

    def forward(self, x):
        x = torch.relu(self.stem_bn(self.stem(x)))
        for b in self.blocks:
            x = b(x)
        policy = torch.flatten(self.policy(x), 1)
        v = torch.relu(self.value_bn(self.value_conv(x)))
        h = torch.relu(self.fc1(torch.flatten(v, 1)))
        if self.swapped_squashing:  # sigmoid value head, tanh draw head: the same functions, the other way round
            value = torch.sigmoid(self.fc_value(h))
            draw = (torch.tanh(self.fc_draw(h)) + 1.0) * 0.5
        else:
            value = (torch.tanh(self.fc_value(h)) + 1.0) / 2.0
            draw = torch.sigmoid(self.fc_draw(h))
        return policy, value, draw


def build(blocks=BLOCKS, seed=20240203, swapped_squashing=False, eps=None):
    torch.manual_seed(seed)
    net = Net(blocks, swapped_squashing)
    g = torch.Generator().manual_seed(7)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            n = m.num_features
            m.weight.data = torch.rand(n, generator=g) + 0.5
            m.bias.data = torch.randn(n, generator=g) * 0.1
            m.running_mean = torch.randn(n, generator=g) * 0.1
            m.running_var = torch.rand(n, generator=g) + 0.5
            if eps is not None:
                m.eps = eps
    return net.eval()


def export(net, path, fold):
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils
    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto  # see the module docstring
    buf = io.BytesIO()
    torch.onnx.export(net, (torch.zeros(1, C, 9, 9),), buf, input_names=["input"],
                      output_names=["policy", "value", "draw"], dynamo=False, opset_version=17,
                      dynamic_axes={"input": {0: "N"}, "policy": {0: "N"}, "value": {0: "N"}, "draw": {0: "N"}},
                      do_constant_folding=fold,
                      training=torch.onnx.TrainingMode.EVAL if fold else torch.onnx.TrainingMode.PRESERVE)
    with open(path, "wb") as f:
        f.write(buf.getvalue())
    return buf.getvalue()


def main():
    import importlib
    import shogi_ref
    import test_features
    nsg = importlib.import_module("nshogi-engine_amd")
    here = os.path.join(ROOT, "tests", "golden")
    rows = test_features.dump(1, 20240203, 1024, 0.5, 60)
    bb = np.stack([r[1] for r in rows[::12]])[:6]
    planes = nsg.synth.expand_reference(bb, True).reshape(-1, C, 9, 9)
    out = {"bitboards": bb}
    # (a) the usual export: eval mode, constant folding -> BatchNorm folded into conv weight + bias
    # (b) BatchNormalization nodes kept (training=PRESERVE, no folding), one block
    # (c) sigmoid value head + tanh draw head, BatchNormalization nodes kept with epsilon 1e-3 (not the default):
    #     the policy conv (bias, no BN) must still fold with scale exactly 1
    for name, blocks, fold, seed, swapped, eps in (("net_torch_2x64", 2, True, 20240203, False, None),
                                                   ("net_torch_bn_1x64", 1, False, 20240204, False, None),
                                                   ("net_torch_sigtanh_eps_1x64", 1, False, 20240205, True, 1e-3)):
        net = build(blocks, seed, swapped, eps)
        data = export(net, os.path.join(here, name + ".onnx"), fold)
        with torch.no_grad():
            p, v, d = net.double()(torch.from_numpy(planes).double())
        out[name + "_policy"] = p.numpy()
        out[name + "_value"] = v.numpy().reshape(-1)
        out[name + "_draw"] = d.numpy().reshape(-1)
        print(name, "onnx bytes", len(data), "positions", len(bb), "torch", torch.__version__)
    np.savez_compressed(os.path.join(here, "net_torch.npz"), **out)


if __name__ == "__main__":
    main()
