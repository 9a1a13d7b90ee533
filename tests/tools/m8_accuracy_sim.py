#!/usr/bin/env python3
"""CPU simulation (float64 torch) of the "f16m8" arithmetic: main term w_hi*x_hi on the f16
MFMA, correction terms w_lo*x_hi and w_hi*x_lo on the fp8 (e4m3) MX MFMA with FIXED
power-of-two scales, activations stored as (f16 hi, fp8(x_hi), fp8(x_lo * 2^SL)) so the
residual stream carries hi + dequantised lo.  Prints max/rms error of the policy logits
against float64 for f16, f16x3 and f16m8 on the synthetic 20x256 net."""
import importlib, sys, numpy as np, torch, torch.nn.functional as F
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
nsg = importlib.import_module('nshogi-engine_amd'); import oracle_lib
torch.set_num_threads(8)
SL, SW, SH = 12, 10, -2
def q_f16(t): return t.to(torch.float16).to(torch.float64)
def q_e4m3(t):  # round to nearest fp8 e4m3 (max 448, subnormal step 2^-9), saturating
    a = t.abs().clamp_min(1e-300)
    e = torch.floor(torch.log2(a)).clamp(min=-6)
    q = torch.round(t / torch.pow(2.0, e - 3)) * torch.pow(2.0, e - 3)
    return q.clamp(-448, 448)
def q8s(t, s): return q_e4m3(t * 2.0**s) / 2.0**s
def split_store(v):
    hi = q_f16(v); return hi, q8s(v - hi, SL)
def conv(xh, xl, w, mode, pad):
    if mode == 'f64': return F.conv2d(xh + xl, w, padding=pad)
    e = torch.floor(torch.log2(w.abs().max())); s = 2.0 ** (9 - e)
    ws = w * s; wh = q_f16(ws); wl = ws - wh
    if mode == 'f16': return F.conv2d(xh, wh, padding=pad) / s
    if mode == 'f16x3': return (F.conv2d(xh, wh, padding=pad) + F.conv2d(xh, q_f16(wl), padding=pad) + F.conv2d(xl, wh, padding=pad)) / s
    return (F.conv2d(xh, wh, padding=pad) + F.conv2d(q_e4m3(xh), q8s(wl, SW), padding=pad) + F.conv2d(xl, q8s(wh, SH), padding=pad)) / s
def forward(w, planes, mode):
    m = w['_meta']; eps = m['bn_eps']; t = lambda a: torch.from_numpy(np.asarray(a)).double()
    def fold(wt, bn):
        g, b, mu, var = [t(bn[i]) for i in range(4)]; s = g / torch.sqrt(var + eps)
        return t(wt) * s.view(-1, 1, 1, 1), b - mu * s
    def store(v):
        if mode == 'f64': return v, torch.zeros_like(v)
        hi = q_f16(v)
        if mode == 'f16': return hi, torch.zeros_like(v)
        if mode == 'f16x3': return hi, q_f16(v - hi)
        return hi, q8s(v - hi, SL)
    xh, xl = store(t(planes).view(-1, m['in_channels'], 9, 9))
    amax = 0.0
    W, B = fold(w['stem_w'], w['stem_bn']); xh, xl = store(F.relu(conv(xh, xl, W, mode, 1) + B.view(1, -1, 1, 1)))
    for k in range(m['blocks']):
        W, B = fold(w[f'b{k}_w1'], w[f'b{k}_bn1']); yh, yl = store(F.relu(conv(xh, xl, W, mode, 1) + B.view(1, -1, 1, 1)))
        W, B = fold(w[f'b{k}_w2'], w[f'b{k}_bn2']); xh, xl = store(F.relu(xh + xl + conv(yh, yl, W, mode, 1) + B.view(1, -1, 1, 1)))
        amax = max(amax, float(xh.abs().max()), float(yh.abs().max()))
    Fc = m['channels']
    pol = conv(xh, xl, t(w['policy_w']).view(27, Fc, 1, 1), mode, 0) + t(w['policy_b']).view(1, -1, 1, 1)
    return pol.reshape(-1, 2187).numpy(), amax
for blocks, ch, bn in ((20, 256, 'identity'), (20, 256, 'random'), (10, 192, 'random')):
    w = nsg.weights.make_random(blocks, ch, seed=0, bn=bn)
    bb = nsg.synth.random_batch(2, 86, seed=1)
    planes = oracle_lib.load().extract_bits(bb)
    ref, _ = forward(w, planes, 'f64')
    for mode in ('f16', 'f16x3', 'f16m8'):
        out, amax = forward(w, planes, mode)
        print(f'{blocks}x{ch} bn={bn} {mode:6s} max abs err {np.abs(out - ref).max():.3e} rms {np.sqrt(((out-ref)**2).mean()):.3e}  (logit range {np.abs(ref).max():.2f}, act max {amax:.1f})', flush=True)
