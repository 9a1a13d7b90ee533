"""Parity at the BENCHMARK's arithmetic on the positions that are actually evaluated: the initial
position (the reference benchmark's input in every slot, /root/reference/src/bench/batchsize.cc:47-59)
and positions of real games (what self-play feeds the evaluator) -- saturated hand / scalar planes,
sparse boards -- not the seeded random bitboards of the other parity tests.  A sample of >= 32 boards,
stratified over the plies of the games, against the CPU oracle at the north_star's 1e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3  # north_star: "within 1e-3 fp32"


def _batch(nsg, batch):
    """Half the slots hold the initial position, the other half distinct positions of random-playout
    games in game order (ply 0, 1, 2, ... of game after game)."""
    half = batch // 2
    return np.ascontiguousarray(np.concatenate([nsg.positions.startpos_batch(half),
                                                nsg.positions.game_positions(batch - half, seed=20240203)]))


@pytest.mark.parametrize("blocks,channels,batch,precision,sample", [
    (20, 256, 512, "f16m6", 40),   # bench.py's line: BASELINE configs[2]
    (20, 256, 512, "f16m8", 32),
    (20, 256, 128, "f16m6", 32),   # the batch self-play runs at (K-split tiles)
    (10, 192, 64, "f16m6", 32),    # BASELINE configs[1]
])
def test_real_positions_at_the_benchmark_arithmetic(nsg, oracle, blocks, channels, batch, precision, sample):
    blob = nsg.weights.to_blob(nsg.weights.make_random(blocks, channels, seed=0, bn="identity"))  # bench.py's weights
    ev = nsg.Evaluator(0, batch, 86, precision=precision)
    ev.load_memory(blob)
    bb = _batch(nsg, batch)
    p, v, d = ev.compute_blocking(bb)
    half = batch // 2
    # every slot holding the initial position returns the same bits (batchsize.cc:52-59 replicates it)
    assert (p[:half] == p[0]).all() and (v[:half] == v[0]).all() and (d[:half] == d[0]).all()
    # the initial position + a stratified sample of the game positions (evenly spaced over the plies)
    idx = np.unique(np.concatenate([[0], np.linspace(half, batch - 1, sample - 1).round().astype(int)]))
    assert len(idx) >= 32
    po, vo, do = oracle.net(blob).evaluate_parallel(bb[idx])
    err = (float(np.abs(p[idx] - po).max()), float(np.abs(v[idx] - vo).max()), float(np.abs(d[idx] - do).max()))
    print(f"{blocks}x{channels} B={batch} {precision}: max|err| policy {err[0]:.2e} value {err[1]:.2e} draw {err[2]:.2e} "
          f"on {len(idx)} boards (logit range {np.abs(po).max():.2f})")
    assert max(err) <= TOL, err
    assert np.isfinite(p).all() and ((v >= 0) & (v <= 1)).all() and ((d >= 0) & (d <= 1)).all()
    ev.close()


def _trained_like_weights(nsg, blocks, channels, steps, seed):
    """Weights with the statistics of a net that has been TRAINED, not drawn: a torch module of the topology of
    DESIGN.md section 2 takes `steps` SGD steps on real game positions (BatchNorm in training mode: the running
    means / variances become those of its own activations; the convolutions leave their He-normal start), then a
    function-preserving re-parametrisation gives every channel of the residual stream and of the block interiors
    its own scale, log-uniform over 1/16 .. 16 (gamma and beta of the producing BatchNorm times s, the consuming
    convolution's input weights divided by s; ReLU commutes with a positive scale) -- the heavy-tailed per-channel
    magnitudes of trained nets that one exponent per 32 channels and the +-65000 range of the f16 copies have to
    live with.  Returns the NSGW weight dict and the module (eval mode, float64)."""
    import torch
    import torch.nn as nn
    torch.manual_seed(seed)
    F, C, VC, VH = channels, 86, 32, 256

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(F, F, 3, padding=1, bias=False), nn.BatchNorm2d(F)
            self.conv2, self.bn2 = nn.Conv2d(F, F, 3, padding=1, bias=False), nn.BatchNorm2d(F)

        def forward(self, x):
            return torch.relu(x + self.bn2(self.conv2(torch.relu(self.bn1(self.conv1(x))))))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem, self.stem_bn = nn.Conv2d(C, F, 3, padding=1, bias=False), nn.BatchNorm2d(F)
            self.blocks = nn.ModuleList([Block() for _ in range(blocks)])
            self.policy = nn.Conv2d(F, 27, 1)
            self.value_conv, self.value_bn = nn.Conv2d(F, VC, 1, bias=False), nn.BatchNorm2d(VC)
            self.fc1, self.fc2 = nn.Linear(VC * 81, VH), nn.Linear(VH, 2)

        def forward(self, x):
            x = torch.relu(self.stem_bn(self.stem(x)))
            for b in self.blocks:
                x = b(x)
            v = torch.relu(self.value_bn(self.value_conv(x)))
            o = self.fc2(torch.relu(self.fc1(torch.flatten(v, 1))))
            return torch.flatten(self.policy(x), 1), (torch.tanh(o[:, 0]) + 1) / 2, torch.sigmoid(o[:, 1])

    net = Net()
    bb = nsg.positions.game_positions(32 * steps, seed=seed)
    planes = torch.from_numpy(nsg.synth.expand_reference(bb, True).reshape(-1, C, 9, 9).astype(np.float32))
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    g = torch.Generator().manual_seed(seed)
    net.train()
    for k in range(steps):
        x = planes[32 * k:32 * k + 32]
        tgt = torch.randint(0, 2187, (32,), generator=g)
        out = torch.rand(32, generator=g)
        p, v, d = net(x)
        loss = nn.functional.cross_entropy(p, tgt) + ((v - out) ** 2).mean() + ((d - 0.1) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    net.eval()
    rng = np.random.default_rng(seed)

    def scales():
        return torch.from_numpy(np.exp2(rng.uniform(-4, 4, F)).astype(np.float32))

    with torch.no_grad():
        s = scales()  # the residual stream: the stem's and every block's second BatchNorm produce it
        for bn in [net.stem_bn] + [b.bn2 for b in net.blocks]:
            bn.weight.mul_(s)
            bn.bias.mul_(s)
        for conv in [b.conv1 for b in net.blocks] + [net.policy, net.value_conv]:
            conv.weight.div_(s.view(1, F, 1, 1))
        for b in net.blocks:  # block interiors
            t = scales()
            b.bn1.weight.mul_(t)
            b.bn1.bias.mul_(t)
            b.conv2.weight.div_(t.view(1, F, 1, 1))

    def bn4(m):
        return np.stack([m.weight.detach().numpy(), m.bias.detach().numpy(), m.running_mean.numpy(),
                         m.running_var.numpy()]).astype(np.float32)

    w = {"stem_w": net.stem.weight.detach().numpy(), "stem_bn": bn4(net.stem_bn)}
    for k, b in enumerate(net.blocks):
        w[f"b{k}_w1"], w[f"b{k}_bn1"] = b.conv1.weight.detach().numpy(), bn4(b.bn1)
        w[f"b{k}_w2"], w[f"b{k}_bn2"] = b.conv2.weight.detach().numpy(), bn4(b.bn2)
    w.update({"policy_w": net.policy.weight.detach().numpy().reshape(27, F), "policy_b": net.policy.bias.detach().numpy(),
              "value_w": net.value_conv.weight.detach().numpy().reshape(VC, F), "value_bn": bn4(net.value_bn),
              "fc1_w": net.fc1.weight.detach().numpy(), "fc1_b": net.fc1.bias.detach().numpy(),
              "fc2_w": net.fc2.weight.detach().numpy(), "fc2_b": net.fc2.bias.detach().numpy()})
    w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}
    w["_meta"] = dict(blocks=blocks, channels=F, in_channels=C, policy_channels=27, value_channels=VC, value_hidden=VH,
                      bn_eps=float(net.stem_bn.eps))
    return w, net.double()


@pytest.mark.parametrize("precision,batch", [("f16m6", 512), ("f16m6", 96), ("f16m8", 512), ("f16x3", 40), ("f16m6", 6)])
def test_trained_weight_statistics(nsg, oracle, precision, batch):
    """Every other parity test draws He-normal weights with BatchNorm statistics near (0, 1).  Here the net has
    the statistics of a trained one (see _trained_like_weights): BatchNorm running moments of real activations, and
    per-channel magnitudes spread over 1/16 .. 16 -- within one 32-channel block of the f16m6 format, whose fp6
    copies share one exponent, and up to 16 x the usual range against the f16 clamp.  Real game positions, the tile
    plans of batch 512 (two-board MX tiles), 96 (K split), 40 (f16x3 small tiles) and 6 (team trunk), against the
    oracle at the north_star's 1e-3 and PyTorch's float64 forward of the same module."""
    import torch
    w, net = _trained_like_weights(nsg, 2, 256, 6, 20240203)
    blob = nsg.weights.to_blob(w)
    ev = nsg.Evaluator(0, batch, 86, precision=precision)
    ev.load_memory(blob)
    bb = np.ascontiguousarray(nsg.positions.game_positions(batch, seed=77))
    p, v, d = ev.compute_blocking(bb)
    idx = np.unique(np.linspace(0, batch - 1, min(batch, 32)).round().astype(int))
    po, vo, do = oracle.net(blob).evaluate_parallel(bb[idx])
    with torch.no_grad():
        planes = torch.from_numpy(nsg.synth.expand_reference(bb[idx], True).reshape(-1, 86, 9, 9)).double()
        pt, vt, dt = (t.numpy() for t in net(planes))
    assert float(np.abs(po - pt).max()) < 2e-4 and float(np.abs(vo - vt).max()) < 1e-5  # the oracle agrees with PyTorch
    err = (float(np.abs(p[idx] - po).max()), float(np.abs(v[idx] - vo).max()), float(np.abs(d[idx] - do).max()))
    info = ev.info()
    print(f"trained-like 2x256 B={batch} {precision} ({ev.last_plan()['trunk_precision']}): max|err| policy {err[0]:.2e} "
          f"value {err[1]:.2e} draw {err[2]:.2e}; logit range {np.abs(po).max():.1f}, activation bound estimate "
          f"{info['activation_bound_estimate']:.0f}, f16m8 window fallback {info['f16m8_window_fallback']}")
    assert max(err) <= TOL, err
    assert np.isfinite(p).all() and ((v >= 0) & (v <= 1)).all() and ((d >= 0) & (d <= 1)).all()
    ev.close()
