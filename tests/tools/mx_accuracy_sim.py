#!/usr/bin/env python3
"""CPU simulation (float64 torch) of candidate arithmetic for the trunk: plain f16, the
shipped f16x3 split, and "f16 main term + block-scaled MX minifloat correction terms"
(x*w ~ x_hi*w_hi + Q(x_lo)*Q(w) + Q(x)*Q(w_lo) with Q = MX fp8/fp6/fp4, the formats of
gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 which run 2x / 4x / 4x the f16 rate).
Measured on the 20x256 synthetic net (max abs error of the policy logits vs float64):
f16 2.9e-3, f16x3 1.6e-5, MX-fp8 1.0e-4, MX-fp6 1.2e-4, MX-fp4 4.9e-4 -- i.e. fp6
corrections would keep an 8x margin under the 1e-3 bar at half the matrix-pipe cycles
of f16x3.  Not implemented this round; the operand layout of the scaled
MFMA was probed in scripts/probe/mx_probe.hip."""
import importlib, sys, numpy as np, torch, torch.nn.functional as F
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
nsg = importlib.import_module('nshogi-engine_amd'); import oracle_lib
torch.set_num_threads(8)
def q_f16(t): return t.to(torch.float16).to(torch.float64)
def q_mx(t, dim, mant_bits, emax, emin):
    # block-scaled minifloat along `dim` in blocks of 32: shared power-of-two scale, element = minifloat
    t = t.movedim(dim, -1); sh = t.shape; C = sh[-1]; pad = (-C) % 32
    if pad: t = F.pad(t, (0, pad))
    b = t.reshape(*t.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    s = torch.floor(torch.log2(amax)) - emax
    v = b / torch.pow(2.0, s)
    e = torch.floor(torch.log2(v.abs().clamp_min(1e-300))).clamp(min=emin)
    q = torch.round(v / torch.pow(2.0, e - mant_bits)) * torch.pow(2.0, e - mant_bits)
    lim = (2 - 2.0**-mant_bits) * 2.0**emax
    q = q.clamp(-lim, lim) * torch.pow(2.0, s)
    q = q.reshape(*t.shape)[..., :C].reshape(sh)
    return q.movedim(-1, dim)
def conv(x, w, mode):
    if mode == 'f64': return F.conv2d(x, w, padding=w.shape[-1]//2)
    xh, wh = q_f16(x), q_f16(w); xl, wl = x - xh, w - wh
    if mode == 'f16': return F.conv2d(xh, wh, padding=w.shape[-1]//2)
    if mode == 'f16x3': return F.conv2d(xh, wh, padding=w.shape[-1]//2) + F.conv2d(q_f16(xl), wh, padding=w.shape[-1]//2) + F.conv2d(xh, q_f16(wl), padding=w.shape[-1]//2)
    mb, emax, emin = {'fp8': (3, 8, -6), 'fp6': (3, 2, 0), 'fp4': (1, 2, 0)}[mode]
    xlq = q_mx(xl, 1, mb, emax, emin); wlq = q_mx(wl, 1, mb, emax, emin)
    xq = q_mx(x, 1, mb, emax, emin); wq = q_mx(w, 1, mb, emax, emin)   # both factors of a correction product are MX minifloats
    return F.conv2d(xh, wh, padding=w.shape[-1]//2) + F.conv2d(xlq, wq, padding=w.shape[-1]//2) + F.conv2d(xq, wlq, padding=w.shape[-1]//2)
def forward(w, planes, mode):
    m = w['_meta']; eps = m['bn_eps']; t = lambda a: torch.from_numpy(np.asarray(a)).double()
    def fold(wt, bn):
        g, b, mu, var = [t(bn[i]) for i in range(4)]; s = g / torch.sqrt(var + eps)
        return t(wt) * s.view(-1, 1, 1, 1), b - mu * s
    x = t(planes).view(-1, m['in_channels'], 9, 9)
    W, B = fold(w['stem_w'], w['stem_bn']); x = F.relu(conv(x, W, mode) + B.view(1, -1, 1, 1))
    for k in range(m['blocks']):
        W, B = fold(w[f'b{k}_w1'], w[f'b{k}_bn1']); y = F.relu(conv(x, W, mode) + B.view(1, -1, 1, 1))
        W, B = fold(w[f'b{k}_w2'], w[f'b{k}_bn2']); x = F.relu(x + conv(y, W, mode) + B.view(1, -1, 1, 1))
    Fc = m['channels']
    pol = conv(x, t(w['policy_w']).view(27, Fc, 1, 1), mode) + t(w['policy_b']).view(1, -1, 1, 1)
    return pol.reshape(-1, 2187).numpy()
blocks, ch = 20, 256
w = nsg.weights.make_random(blocks, ch, seed=0)
bb = nsg.synth.random_batch(2, 86, seed=1)
planes = oracle_lib.load().extract_bits(bb)
ref = forward(w, planes, 'f64')
for mode in ('f16', 'f16x3', 'fp8', 'fp6', 'fp4'):
    out = forward(w, planes, mode)
    print(mode, 'max abs err', float(np.abs(out - ref).max()), 'rms', float(np.sqrt(((out-ref)**2).mean())))
