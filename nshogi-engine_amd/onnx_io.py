"""ONNX <-> NSGW bridge without the onnx / protobuf packages (neither is in this image).

The reference loads whatever ONNX file it is given through TensorRT's parser
(src/infer/trt.cc:109-131) and fixes only the tensor contract: input "input"
[N,C,9,9], outputs "policy" (2187 values per position), "value", "draw"
(trt.cc:144-150,193-227).  This module

  * writes a weight dict of this build's topology (weights.py) as an ONNX
    model obeying that contract (`export_onnx`) -- so the same network can be run through the
    reference's TensorRT executor for a side-by-side check, and
  * reads an ONNX model of that topology family back into a weight dict (`import_onnx`):
    stem conv3x3 [+BN] +ReLU, N x (conv-BN-ReLU-conv-BN-add-ReLU), 1x1 policy conv,
    1x1 value conv + BN + ReLU + two dense layers, value = (tanh+1)/2 or sigmoid,
    draw = sigmoid.  Anything else is refused with a message naming the node.

The protobuf wire format is hand-coded from the public onnx.proto3 field numbers.  The READER
is pinned against models serialised by PyTorch's own ONNX exporter (tests/golden/net_torch_*.onnx,
made by tests/golden/make_onnx_golden.py); the product-side reader is the C++ one inside
libnsg.so (csrc/onnx_reader.cc, used by nsg_load) and this module is its independent cross-check
in the tests.  The WRITER has only been read back by these two readers (no onnxruntime / TensorRT
here): unverified against the real consumers.
"""
import struct

import numpy as np

from . import weights as _weights

# ---- protobuf wire format ---------------------------------------------------------------


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode("utf-8")
    return _key(field, 2) + _varint(len(b)) + bytes(b)


def _f_float(field, v):
    return _key(field, 5) + struct.pack("<f", float(v))


def _parse(buf):
    """Yields (field, wire_type, value) of one message; length-delimited values are memoryviews."""
    mv = memoryview(buf)
    i, n = 0, len(mv)
    while i < n:
        k, shift = 0, 0
        while True:
            b = mv[i]
            i += 1
            k |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, shift = 0, 0
            while True:
                b = mv[i]
                i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield field, wire, v
        elif wire == 1:
            yield field, wire, bytes(mv[i:i + 8])
            i += 8
        elif wire == 2:
            ln, shift = 0, 0
            while True:
                b = mv[i]
                i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield field, wire, mv[i:i + ln]
            i += ln
        elif wire == 5:
            yield field, wire, bytes(mv[i:i + 4])
            i += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- ONNX messages (onnx.proto3 field numbers) ---------------------------------------------
_FLOAT, _INT64 = 1, 7
_AT_FLOAT, _AT_INT, _AT_INTS = 1, 2, 7


def _tensor(name, arr):
    arr = np.ascontiguousarray(arr)
    dt = _FLOAT if arr.dtype == np.float32 else _INT64
    out = b"".join(_f_varint(1, d) for d in arr.shape)
    out += _f_varint(2, dt) + _f_bytes(8, name) + _f_bytes(9, arr.tobytes())
    return out


def _attr_i(name, v):
    return _f_bytes(1, name) + _f_varint(3, v) + _f_varint(20, _AT_INT)


def _attr_f(name, v):
    return _f_bytes(1, name) + _f_float(2, v) + _f_varint(20, _AT_FLOAT)


def _attr_ints(name, vs):
    return _f_bytes(1, name) + b"".join(_f_varint(8, v) for v in vs) + _f_varint(20, _AT_INTS)


def _node(op, inputs, outputs, attrs=(), name=None):
    out = b"".join(_f_bytes(1, i) for i in inputs) + b"".join(_f_bytes(2, o) for o in outputs)
    out += _f_bytes(3, name or outputs[0]) + _f_bytes(4, op)
    out += b"".join(_f_bytes(5, a) for a in attrs)
    return out


def _value_info(name, dims):
    shape = b""
    for d in dims:
        shape += _f_bytes(1, _f_bytes(2, d) if isinstance(d, str) else _f_varint(1, d))
    ttype = _f_varint(1, _FLOAT) + _f_bytes(2, shape)
    return _f_bytes(1, name) + _f_bytes(2, _f_bytes(1, ttype))


def export_onnx(w, path=None, opset=13, fold_bn=False, value_sigmoid=False):
    """Serialises a weight dict (weights.make_random / from_blob) as an ONNX model.
    fold_bn: fold every BatchNormalization into its convolution (weights scaled in float64, a
    conv bias instead of the BN node); value_sigmoid: emit value = sigmoid(2 * logit), the same
    function as (tanh(logit) + 1) / 2, as one node."""
    m = w["_meta"]
    blocks, F, C = m["blocks"], m["channels"], m["in_channels"]
    VC, VH, PC, eps = m["value_channels"], m["value_hidden"], m["policy_channels"], m["bn_eps"]
    inits, nodes = [], []

    def init(name, arr):
        inits.append(_tensor(name, np.asarray(arr)))
        return name

    def conv(x, wname, wt, y, bias=None, k=3):
        ins = [x, init(wname, np.asarray(wt, np.float32))]
        if bias is not None:
            ins.append(init(wname + "_bias", np.asarray(bias, np.float32)))
        p = k // 2
        nodes.append(_node("Conv", ins, [y], [_attr_ints("kernel_shape", [k, k]), _attr_ints("pads", [p, p, p, p]),
                                              _attr_ints("strides", [1, 1]), _attr_ints("dilations", [1, 1]),
                                              _attr_i("group", 1)]))
        return y

    def conv_bn(x, wname, wt, stats, y, k=3):
        if not fold_bn:
            return bn(conv(x, wname, wt, y + "_conv", k=k), wname + "_bn", stats, y)
        g, b, mu, var = (np.asarray(stats[i], np.float64) for i in range(4))
        sc = g / np.sqrt(var + eps)
        wf = (np.asarray(wt, np.float64) * sc.reshape(-1, 1, 1, 1)).astype(np.float32)
        return conv(x, wname, wf, y, bias=(b - mu * sc).astype(np.float32), k=k)

    def bn(x, prefix, stats, y):
        g, b, mu, var = (np.asarray(stats[i], np.float32) for i in range(4))
        nodes.append(_node("BatchNormalization",
                           [x, init(prefix + "_gamma", g), init(prefix + "_beta", b), init(prefix + "_mean", mu),
                            init(prefix + "_var", var)], [y], [_attr_f("epsilon", eps)]))
        return y

    def unary(op, x, y):
        nodes.append(_node(op, [x], [y]))
        return y

    x = unary("Relu", conv_bn("input", "stem_w", w["stem_w"], w["stem_bn"], "stem_out"), "trunk0")
    for k in range(blocks):
        y = unary("Relu", conv_bn(x, f"b{k}_w1", w[f"b{k}_w1"], w[f"b{k}_bn1"], f"b{k}_n1"), f"b{k}_r1")
        y = conv_bn(y, f"b{k}_w2", w[f"b{k}_w2"], w[f"b{k}_bn2"], f"b{k}_n2")
        nodes.append(_node("Add", [x, y], [f"b{k}_sum"]))
        x = unary("Relu", f"b{k}_sum", f"trunk{k + 1}")
    # policy: 1x1 conv, NCHW flatten -> index c*81 + sq
    conv(x, "policy_w", np.asarray(w["policy_w"]).reshape(PC, F, 1, 1), "policy_map", bias=w["policy_b"], k=1)
    nodes.append(_node("Flatten", ["policy_map"], ["policy"], [_attr_i("axis", 1)]))
    # value / draw
    v = unary("Relu", conv_bn(x, "value_w", np.asarray(w["value_w"]).reshape(VC, F, 1, 1), w["value_bn"], "value_bn_out", k=1),
              "value_map")
    nodes.append(_node("Flatten", [v], ["value_flat"], [_attr_i("axis", 1)]))
    gemm = [_attr_f("alpha", 1.0), _attr_f("beta", 1.0), _attr_i("transA", 0), _attr_i("transB", 1)]
    nodes.append(_node("Gemm", ["value_flat", init("fc1_w", w["fc1_w"]), init("fc1_b", w["fc1_b"])], ["fc1_out"], gemm))
    unary("Relu", "fc1_out", "hidden")
    fc2w, fc2b = np.asarray(w["fc2_w"], np.float32), np.asarray(w["fc2_b"], np.float32)
    vs = 2.0 if value_sigmoid else 1.0  # sigmoid(2 z) = (tanh(z) + 1) / 2; doubling an f32 is exact
    nodes.append(_node("Gemm", ["hidden", init("value_fc2_w", fc2w[0:1] * vs), init("value_fc2_b", fc2b[0:1] * vs)], ["value_logit"], gemm))
    nodes.append(_node("Gemm", ["hidden", init("draw_fc2_w", fc2w[1:2]), init("draw_fc2_b", fc2b[1:2])], ["draw_logit"], gemm))
    if value_sigmoid:
        unary("Sigmoid", "value_logit", "value")
    else:
        unary("Tanh", "value_logit", "value_tanh")
        nodes.append(_node("Add", ["value_tanh", init("one", np.array([1.0], np.float32))], ["value_shift"]))
        nodes.append(_node("Mul", ["value_shift", init("half", np.array([0.5], np.float32))], ["value"]))
    unary("Sigmoid", "draw_logit", "draw")

    graph = b"".join(_f_bytes(1, n) for n in nodes) + _f_bytes(2, "nshogi_policy_value_draw")
    graph += b"".join(_f_bytes(5, t) for t in inits)
    graph += _f_bytes(11, _value_info("input", ["N", C, 9, 9]))
    graph += _f_bytes(12, _value_info("policy", ["N", PC * 81]))
    graph += _f_bytes(12, _value_info("value", ["N", 1])) + _f_bytes(12, _value_info("draw", ["N", 1]))
    model = _f_varint(1, 7) + _f_bytes(2, "nshogi-engine_amd") + _f_bytes(7, graph)
    model += _f_bytes(8, _f_bytes(1, "") + _f_varint(2, opset))
    if path is not None:
        with open(path, "wb") as f:
            f.write(model)
    return model


# ---- reader --------------------------------------------------------------------------------


class _Node:
    __slots__ = ("op", "inputs", "outputs", "name", "attrs")


def _read_tensor(mv):
    dims, dtype, name, raw, floats, int64s = [], None, "", None, [], []
    for f, wt, v in _parse(mv):
        if f == 1:
            if wt == 0:
                dims.append(v)
            else:  # packed
                dims += [x for _, _, x in _parse_packed_varints(v)]
        elif f == 2:
            dtype = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            if wt == 5:
                floats.append(struct.unpack("<f", v)[0])
            else:
                floats += list(np.frombuffer(bytes(v), "<f4"))
        elif f == 7:
            if wt == 0:
                int64s.append(_signed(v))
            else:
                int64s += [_signed(x) for _, _, x in _parse_packed_varints(v)]
    if dtype == _FLOAT:
        arr = np.frombuffer(raw, "<f4") if raw is not None else np.asarray(floats, np.float32)
    elif dtype == _INT64:
        arr = np.frombuffer(raw, "<i8") if raw is not None else np.asarray(int64s, np.int64)
    else:
        raise ValueError(f"initializer {name!r}: unsupported data type {dtype} (float32 / int64 only)")
    return name, np.array(arr).reshape(dims)


def _parse_packed_varints(mv):
    i, n = 0, len(mv)
    while i < n:
        v, shift = 0, 0
        while True:
            b = mv[i]
            i += 1
            v |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        yield 0, 0, v


def _read_node(mv):
    nd = _Node()
    nd.inputs, nd.outputs, nd.name, nd.op, nd.attrs = [], [], "", "", {}
    for f, wt, v in _parse(mv):
        if f == 1:
            nd.inputs.append(bytes(v).decode())
        elif f == 2:
            nd.outputs.append(bytes(v).decode())
        elif f == 3:
            nd.name = bytes(v).decode()
        elif f == 4:
            nd.op = bytes(v).decode()
        elif f == 5:
            name, val, ints = "", None, []
            for af, awt, av in _parse(v):
                if af == 1:
                    name = bytes(av).decode()
                elif af == 2:
                    val = struct.unpack("<f", av)[0]
                elif af == 3:
                    val = _signed(av)
                elif af == 8:
                    if awt == 0:
                        ints.append(_signed(av))
                    else:
                        ints += [_signed(x) for _, _, x in _parse_packed_varints(av)]
                elif af == 5:
                    val = _read_tensor(av)[1]
            nd.attrs[name] = ints if ints else val
    return nd


def read_onnx(data):
    """Minimal ModelProto reader: (nodes, initializers{name: array}, inputs, outputs)."""
    graph = None
    for f, wt, v in _parse(data):
        if f == 7:
            graph = v
    if graph is None:
        raise ValueError("not an ONNX ModelProto: no graph")
    nodes, inits, ins, outs = [], {}, [], []
    for f, wt, v in _parse(graph):
        if f == 1:
            nodes.append(_read_node(v))
        elif f == 5:
            name, arr = _read_tensor(v)
            inits[name] = arr
        elif f in (11, 12):
            for vf, _, vv in _parse(v):
                if vf == 1:
                    (ins if f == 11 else outs).append(bytes(vv).decode())
    for nd in nodes:  # Constant nodes are initializers in all but name (torch emits them for scalar literals)
        if nd.op == "Constant" and len(nd.outputs) == 1:
            v = nd.attrs.get("value", nd.attrs.get("value_float"))
            if v is not None:
                inits[nd.outputs[0]] = np.asarray(v, np.float32) if not isinstance(v, np.ndarray) else v
    return nodes, inits, [i for i in ins if i not in inits], outs


def import_onnx(data, bn_eps_default=1e-5):
    """ONNX model (bytes or path) of the topology family above -> weight dict (weights.to_blob)."""
    if isinstance(data, str):
        with open(data, "rb") as f:
            data = f.read()
    nodes, inits, ins, outs = read_onnx(data)
    if "input" not in ins or not {"policy", "value", "draw"} <= set(outs):
        raise ValueError(f"tensor contract (trt.cc:144-227): need input 'input' and outputs policy/value/draw, got {ins} -> {outs}")
    consumers = {}
    for nd in nodes:
        for i in nd.inputs:
            consumers.setdefault(i, []).append(nd)

    def fail(nd, why):
        raise ValueError(f"unsupported ONNX structure at node {nd.name!r} ({nd.op}): {why}")

    def only(t, what):
        c = consumers.get(t, [])
        if len(c) != 1:
            raise ValueError(f"{what}: tensor {t!r} has {len(c)} consumers, expected 1")
        return c[0]

    eps_seen = []
    identity = []  # statistics of convs without BN

    def conv_bn(x, nd, k):
        """Conv [+ BatchNormalization] starting at node nd: returns (weight, bn[4][n], output tensor)."""
        if nd.op != "Conv" or nd.inputs[0] != x:
            fail(nd, f"expected a Conv of {x!r}")
        wt = inits.get(nd.inputs[1])
        if wt is None or wt.ndim != 4 or wt.shape[2:] != (k, k):
            fail(nd, f"expected a constant {k}x{k} weight")
        if nd.attrs.get("group", 1) != 1 or any(s != 1 for s in nd.attrs.get("strides", [1, 1])) or \
                any(d != 1 for d in nd.attrs.get("dilations", [1, 1])) or \
                list(nd.attrs.get("pads", [k // 2] * 4)) != [k // 2] * 4:
            fail(nd, "only stride 1, dilation 1, group 1, 'same' padding")
        n = wt.shape[0]
        bias = inits[nd.inputs[2]] if len(nd.inputs) > 2 else np.zeros(n, np.float32)
        y = nd.outputs[0]
        nxt = consumers.get(y, [])
        if len(nxt) == 1 and nxt[0].op == "BatchNormalization":
            b = nxt[0]
            g, beta, mu, var = (inits[b.inputs[i]].astype(np.float32) for i in range(1, 5))
            eps_seen.append(float(b.attrs.get("epsilon", bn_eps_default)))
            stats = np.stack([g, beta, mu - bias, var])  # a conv bias folds into the BN mean
            return wt.astype(np.float32), stats, b.outputs[0]
        # no BN: identity statistics carrying the conv bias (var + eps == 1 makes the folded scale exactly 1)
        # (the variance row is rewritten below with the epsilon the model's BatchNormalization nodes use)
        stats = np.stack([np.ones(n), bias, np.zeros(n), np.full(n, 1.0 - bn_eps_default)]).astype(np.float32)
        identity.append(stats)
        return wt.astype(np.float32), stats, y

    def relu(t):
        nd = only(t, "activation")
        if nd.op != "Relu":
            fail(nd, "expected Relu")
        return nd.outputs[0]

    w = {}
    first = only("input", "stem")
    w["stem_w"], w["stem_bn"], t = conv_bn("input", first, 3)
    x = relu(t)
    k = 0
    while True:
        cs = consumers.get(x, [])
        convs3 = [c for c in cs if c.op == "Conv" and inits.get(c.inputs[1]) is not None and inits[c.inputs[1]].shape[2:] == (3, 3)]
        adds = [c for c in cs if c.op == "Add"]
        if len(convs3) == 1 and len(adds) == 1 and len(cs) == 2:
            w1, bn1, t = conv_bn(x, convs3[0], 3)
            y = relu(t)
            w2, bn2, t = conv_bn(y, only(y, "second conv of a block"), 3)
            add = only(t, "residual add")
            if add is not adds[0] or set(add.inputs) != {x, t}:
                fail(add, "expected x + conv path")
            x = relu(add.outputs[0])
            w[f"b{k}_w1"], w[f"b{k}_bn1"], w[f"b{k}_w2"], w[f"b{k}_bn2"] = w1, bn1, w2, bn2
            k += 1
            continue
        break
    heads = consumers.get(x, [])
    if len(heads) != 2 or any(h.op != "Conv" for h in heads):
        raise ValueError(f"after {k} residual blocks expected the policy and value 1x1 convs, found {[h.op for h in heads]}")

    def reaches(t, goal):
        seen, todo = set(), [t]
        while todo:
            u = todo.pop()
            if u == goal:
                return True
            for c in consumers.get(u, []):
                for o in c.outputs:
                    if o not in seen:
                        seen.add(o)
                        todo.append(o)
        return False

    pol = [h for h in heads if reaches(h.outputs[0], "policy")]
    val = [h for h in heads if h not in pol]
    if len(pol) != 1 or len(val) != 1:
        raise ValueError("cannot tell the policy head from the value head")
    pw, pbn, t = conv_bn(x, pol[0], 1)
    if not any(pbn is st for st in identity):
        fail(pol[0], "the policy conv must be followed directly by the flatten (no BN)")
    fl = only(t, "policy flatten")
    if fl.op not in ("Flatten", "Reshape") or fl.outputs[0] != "policy":
        fail(fl, "expected Flatten/Reshape -> policy")
    w["policy_w"], w["policy_b"] = pw.reshape(pw.shape[0], pw.shape[1]), (pbn[1] - pbn[2]).astype(np.float32)

    vw, vbn, t = conv_bn(x, val[0], 1)
    t = relu(t)
    fl = only(t, "value flatten")
    if fl.op not in ("Flatten", "Reshape"):
        fail(fl, "expected Flatten/Reshape")
    w["value_w"], w["value_bn"] = vw.reshape(vw.shape[0], vw.shape[1]), vbn

    def dense(t, nd):
        if nd.op == "Gemm" and nd.inputs[0] == t:
            if nd.attrs.get("alpha", 1.0) != 1.0 or nd.attrs.get("beta", 1.0) != 1.0 or nd.attrs.get("transA", 0):
                fail(nd, "Gemm with alpha = beta = 1, transA = 0 only")
            wt = inits[nd.inputs[1]].astype(np.float32)
            if not nd.attrs.get("transB", 0):
                wt = wt.T
            b = inits[nd.inputs[2]].astype(np.float32) if len(nd.inputs) > 2 else np.zeros(wt.shape[0], np.float32)
            return np.ascontiguousarray(wt), b.reshape(-1), nd.outputs[0]
        if nd.op == "MatMul" and nd.inputs[0] == t:
            wt = inits[nd.inputs[1]].astype(np.float32).T
            add = only(nd.outputs[0], "bias add")
            if add.op != "Add":
                fail(add, "expected MatMul + Add")
            b = inits[[i for i in add.inputs if i in inits][0]].astype(np.float32)
            return np.ascontiguousarray(wt), b.reshape(-1), add.outputs[0]
        fail(nd, "expected Gemm or MatMul+Add")

    w["fc1_w"], w["fc1_b"], t = dense(fl.outputs[0], only(fl.outputs[0], "value MLP layer 1"))
    h = relu(t)
    rows = {}
    for nd in consumers.get(h, []):
        wt, b, out = dense(h, nd)
        # follow the squashing to the named output
        cur, scale = out, 1.0
        nd2 = only(cur, "output squashing")
        if nd2.op == "Tanh":  # (tanh(o) + 1) / 2
            a = only(nd2.outputs[0], "tanh shift")
            m2 = only(a.outputs[0], "tanh scale")

            def scalar(node, t):
                other = [i for i in node.inputs if i != t]
                c = inits.get(other[0]) if len(other) == 1 else None
                if c is None or c.size != 1 or (node.op == "Div" and node.inputs[0] != t):
                    fail(node, "expected a scalar float constant operand")
                return float(np.asarray(c).reshape(-1)[0])
            if a.op != "Add" or scalar(a, nd2.outputs[0]) != 1.0:
                fail(a, "expected tanh + 1")
            if not ((m2.op == "Mul" and scalar(m2, a.outputs[0]) == 0.5) or (m2.op == "Div" and scalar(m2, a.outputs[0]) == 2.0)):
                fail(m2, "expected (tanh + 1) * 0.5 or (tanh + 1) / 2")
            name = m2.outputs[0]
            if name == "draw":
                scale = 2.0  # the device evaluates draw as sigmoid(o): (tanh(z) + 1) / 2 = sigmoid(2 z)
        elif nd2.op == "Sigmoid":
            name = nd2.outputs[0]
            if name == "value":
                scale = 0.5  # sigmoid(z) = (tanh(z / 2) + 1) / 2
        else:
            fail(nd2, "expected Tanh or Sigmoid")
        if wt.shape[0] == 1:
            rows[name] = (wt[0] * scale, b[0] * scale)
        else:
            fail(nd, "one dense node per output expected (value, draw)")
    if set(rows) != {"value", "draw"}:
        raise ValueError(f"value MLP outputs found: {sorted(rows)} (need value and draw)")
    w["fc2_w"] = np.stack([rows["value"][0], rows["draw"][0]]).astype(np.float32)
    w["fc2_b"] = np.array([rows["value"][1], rows["draw"][1]], np.float32)

    F = w["stem_w"].shape[0]
    eps = eps_seen[0] if eps_seen else bn_eps_default
    if any(abs(e - eps) > 1e-12 for e in eps_seen):
        raise ValueError("the BatchNormalization nodes use different epsilons; the NSGW header holds one")
    for st in identity:  # a conv without BN folds with scale 1 only if its variance + THIS model's epsilon is 1
        st[3] = np.float32(1.0 - float(np.float32(eps)))
    w["_meta"] = dict(blocks=k, channels=F, in_channels=w["stem_w"].shape[1], policy_channels=w["policy_w"].shape[0],
                      value_channels=w["value_w"].shape[0], value_hidden=w["fc1_w"].shape[0], bn_eps=eps)
    if w["policy_w"].shape[0] * 81 != 2187:
        raise ValueError(f"policy width {w['policy_w'].shape[0] * 81} != 2187 (trt.cc:205)")
    return w


__all__ = ["export_onnx", "import_onnx", "read_onnx"]
_ = _weights  # (weights.to_blob / save turn the imported dict into an .nsgw file)


def main(argv=None):
    """python -m nshogi-engine_amd.onnx_io {import model.onnx out.nsgw | export in.nsgw model.onnx [--fold-bn] [--value-sigmoid]}"""
    import argparse
    ap = argparse.ArgumentParser(prog="onnx_io", description=__doc__.splitlines()[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    imp = sub.add_parser("import", help="ONNX -> NSGW v1")
    imp.add_argument("onnx")
    imp.add_argument("nsgw")
    exp = sub.add_parser("export", help="NSGW v1 -> ONNX")
    exp.add_argument("nsgw")
    exp.add_argument("onnx")
    exp.add_argument("--fold-bn", action="store_true")
    exp.add_argument("--value-sigmoid", action="store_true")
    exp.add_argument("--opset", type=int, default=13)
    a = ap.parse_args(argv)
    if a.cmd == "import":
        w = import_onnx(a.onnx)
        _weights.save(a.nsgw, w)
    else:
        w = _weights.load(a.nsgw)
        export_onnx(w, a.onnx, opset=a.opset, fold_bn=a.fold_bn, value_sigmoid=a.value_sigmoid)
    m = w["_meta"]
    print(f"{a.cmd}: {m['blocks']} blocks x {m['channels']} channels, {m['in_channels']} input planes")


if __name__ == "__main__":
    main()
