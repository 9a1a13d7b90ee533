"""The network oracle (oracle.c, build-defined topology) cross-checked against
an independent float64 PyTorch restatement built from torch.nn.functional ops,
and against the committed golden fixture.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as tf


def torch_forward(w, planes):
    m = w["_meta"]
    eps = m["bn_eps"]
    t = lambda a: torch.from_numpy(np.asarray(a)).double()

    def bn(x, p):
        g, b, mu, var = [t(p[i]).view(1, -1, 1, 1) for i in range(4)]
        return (x - mu) / torch.sqrt(var + eps) * g + b

    x = t(planes).view(-1, m["in_channels"], 9, 9)
    x = tf.relu(bn(tf.conv2d(x, t(w["stem_w"]), padding=1), w["stem_bn"]))
    for k in range(m["blocks"]):
        y = tf.relu(bn(tf.conv2d(x, t(w[f"b{k}_w1"]), padding=1), w[f"b{k}_bn1"]))
        y = bn(tf.conv2d(y, t(w[f"b{k}_w2"]), padding=1), w[f"b{k}_bn2"])
        x = tf.relu(x + y)
    F = m["channels"]
    pol = tf.conv2d(x, t(w["policy_w"]).view(27, F, 1, 1), bias=t(w["policy_b"])).reshape(-1, 2187)
    v = tf.relu(bn(tf.conv2d(x, t(w["value_w"]).view(-1, F, 1, 1)), w["value_bn"]))
    h = tf.relu(tf.linear(v.reshape(v.shape[0], -1), t(w["fc1_w"]), t(w["fc1_b"])))
    o = tf.linear(h, t(w["fc2_w"]), t(w["fc2_b"]))
    value = 0.5 * (torch.tanh(o[:, 0]) + 1.0)
    draw = torch.sigmoid(o[:, 1])
    return pol.numpy(), value.numpy(), draw.numpy(), x.reshape(x.shape[0], F, 81).numpy()


@pytest.mark.parametrize("blocks,channels,bn", [(1, 64, "random"), (3, 64, "random"), (2, 128, "identity")])
def test_oracle_net_vs_torch(nsg, oracle, blocks, channels, bn):
    w = nsg.weights.make_random(blocks, channels, seed=blocks * 10 + channels, bn=bn)
    bb = nsg.synth.random_batch(3, 86, seed=17)
    planes = oracle.extract_bits(bb)
    p, v, d, trunk = oracle.net(nsg.weights.to_blob(w)).forward_planes(planes, want_trunk=True)
    tp, tv, td, tt = torch_forward(w, planes)
    np.testing.assert_allclose(trunk, tt, rtol=0, atol=2e-5)
    np.testing.assert_allclose(p, tp, rtol=0, atol=5e-5)
    np.testing.assert_allclose(v, tv, rtol=0, atol=1e-6)
    np.testing.assert_allclose(d, td, rtol=0, atol=1e-6)
    assert ((v >= 0) & (v <= 1)).all() and ((d >= 0) & (d <= 1)).all()
    assert np.isfinite(p).all()


def test_oracle_net_matches_golden(nsg, oracle, golden_dir):
    g = np.load(f"{golden_dir}/net_tiny.npz")
    w = nsg.weights.make_random(int(g["blocks"]), int(g["channels"]), seed=int(g["weights_seed"]), bn="random")
    p, v, d = oracle.net(nsg.weights.to_blob(w)).evaluate(g["bitboards"])
    np.testing.assert_array_equal(p, g["policy"])
    np.testing.assert_array_equal(v, g["value"])
    np.testing.assert_array_equal(d, g["draw"])


def test_evaluate_is_extract_then_forward(nsg, oracle):
    w = nsg.weights.make_random(1, 64, seed=2)
    net = oracle.net(nsg.weights.to_blob(w))
    bb = nsg.synth.random_batch(2, 86, seed=4)
    a = net.evaluate(bb)
    b = net.forward_planes(oracle.extract_bits(bb))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_blob_roundtrip_and_errors(nsg, oracle):
    w = nsg.weights.make_random(2, 64, seed=1, bn="random")
    blob = nsg.weights.to_blob(w)
    w2 = nsg.weights.from_blob(blob)
    assert all(np.array_equal(w[k], w2[k]) for k in w if k != "_meta")
    assert w2["_meta"]["blocks"] == 2 and w2["_meta"]["channels"] == 64
    with pytest.raises(ValueError):
        oracle.net(blob[:-4])
    with pytest.raises(ValueError):
        oracle.net(b"XXXX" + blob[4:])


def test_flops_formula(nsg):
    # SURVEY.md 8d / BASELINE.md 2
    f = nsg.weights.flops_per_position
    assert abs(f(10, 192) / 1e9 - 1.0999) < 1e-3
    assert abs(f(20, 256) / 1e9 - 3.8553) < 1e-3
    assert abs(f(40, 384) / 1e9 - 17.2491) < 1e-3
