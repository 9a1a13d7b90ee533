#!/usr/bin/env python3
"""CPU simulation (float64 torch) of "f16m6": f16m8 (tests/tools/m8_accuracy_sim.py) with the two
correction products w_lo*x_hi and w_hi*x_lo evaluated on e2m3 (fp6) copies of the operands with one
E8M0 scale per 32 input channels -- per (position, chunk) for the activation copies, per (output
channel, tap, chunk) for the weight copies -- which v_mfma_scale_f32_16x16x128_f8f6f4 applies for
free and retires in HALF the cycles of its e4m3 form (profiles/r02/a_fp6_probe.txt): 1.5 instead
of 2.0 MFMA units per MAC.  Scale rule: the OCP MX one, 2^(floor(log2(max|v|)) - 2), values beyond
7.5 saturate.  Prints max/rms error of the policy logits against float64."""
import importlib, json, sys, numpy as np, torch, torch.nn.functional as F
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
nsg = importlib.import_module('nshogi-engine_amd'); import oracle_lib
torch.set_num_threads(8)
SL, SW, SH = 12, 10, -2
def q_f16(t): return t.to(torch.float16).to(torch.float64)
def q_e4m3(t):
    a = t.abs().clamp_min(1e-300)
    e = torch.floor(torch.log2(a)).clamp(min=-6)
    q = torch.round(t / torch.pow(2.0, e - 3)) * torch.pow(2.0, e - 3)
    return q.clamp(-448, 448)
def q8s(t, s): return q_e4m3(t * 2.0**s) / 2.0**s
def q_e2m3(t):  # round to nearest e2m3: subnormal step 1/8 below 1, 3 mantissa bits, max 7.5, saturating
    a = t.abs().clamp_min(1e-300)
    e = torch.floor(torch.log2(a)).clamp(min=0, max=2)
    step = torch.pow(2.0, e - 3)
    return (torch.round(t / step) * step).clamp(-7.5, 7.5)
def q6_blocks(t, dim):
    """e2m3 with one power-of-two scale per 32 consecutive entries along `dim` (padded with zeros)."""
    t = t.movedim(dim, -1)
    n = t.shape[-1]; pad = (-n) % 32
    tp = F.pad(t, (0, pad)).reshape(*t.shape[:-1], -1, 32)
    m = tp.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(m.clamp_min(2.0 ** -120))) - 2
    s = torch.pow(2.0, e)
    q = (q_e2m3(tp / s) * s).reshape(*t.shape[:-1], -1)[..., :n]
    return q.movedim(-1, dim)
def conv(xh, xl, x6, w, mode, pad):
    if mode == 'f64': return F.conv2d(xh + xl, w, padding=pad)
    e = torch.floor(torch.log2(w.abs().max())); s = 2.0 ** (9 - e)
    ws = w * s; wh = q_f16(ws); wl = ws - wh
    if mode == 'f16': return F.conv2d(xh, wh, padding=pad) / s
    if mode == 'f16x3': return (F.conv2d(xh, wh, padding=pad) + F.conv2d(xh, q_f16(wl), padding=pad) + F.conv2d(xl, wh, padding=pad)) / s
    if mode == 'f16m8': return (F.conv2d(xh, wh, padding=pad) + F.conv2d(q_e4m3(xh), q8s(wl, SW), padding=pad) + F.conv2d(xl, q8s(wh, SH), padding=pad)) / s
    # f16m6: weight copies blocked over the input-channel axis (dim 1), x copies over channels
    return (F.conv2d(xh, wh, padding=pad) + F.conv2d(x6, q6_blocks(wl, 1), padding=pad) + F.conv2d(xl, q6_blocks(wh, 1), padding=pad)) / s
def forward(w, planes, mode, heads_exact=True):
    m = w['_meta']; eps = m['bn_eps']; t = lambda a: torch.from_numpy(np.asarray(a)).double()
    def fold(wt, bn):
        g, b, mu, var = [t(bn[i]) for i in range(4)]; s = g / torch.sqrt(var + eps)
        return t(wt) * s.view(-1, 1, 1, 1), b - mu * s
    def store(v):  # -> hi, lo as the next layer sees it, fp6/fp8 copy of hi
        if mode == 'f64': return v, torch.zeros_like(v), None
        hi = q_f16(v)
        if mode == 'f16': return hi, torch.zeros_like(v), None
        if mode == 'f16x3': return hi, q_f16(v - hi), None
        if mode == 'f16m8': return hi, q8s(v - hi, SL), None
        return hi, q6_blocks(v - hi, 1), q6_blocks(hi, 1)
    xh, xl, x6 = store(t(planes).view(-1, m['in_channels'], 9, 9))
    amax = 0.0
    W, B = fold(w['stem_w'], w['stem_bn']); xh, xl, x6 = store(F.relu(conv(xh, xl, x6, W, mode, 1) + B.view(1, -1, 1, 1)))
    for k in range(m['blocks']):
        W, B = fold(w[f'b{k}_w1'], w[f'b{k}_bn1']); yh, yl, y6 = store(F.relu(conv(xh, xl, x6, W, mode, 1) + B.view(1, -1, 1, 1)))
        W, B = fold(w[f'b{k}_w2'], w[f'b{k}_bn2']); xh, xl, x6 = store(F.relu(xh + xl + conv(yh, yl, y6, W, mode, 1) + B.view(1, -1, 1, 1)))
        amax = max(amax, float(xh.abs().max()), float(yh.abs().max()))
    Fc = m['channels']
    hm = 'f16x3' if mode in ('f16m8', 'f16m6') else mode  # the heads of an f16m8/f16m6 evaluator run as f16x3
    pol = conv(xh, xl, None, t(w['policy_w']).view(27, Fc, 1, 1), hm, 0) + t(w['policy_b']).view(1, -1, 1, 1)
    return pol.reshape(-1, 2187).numpy(), amax
if __name__ == '__main__':
    nets = [(20, 256, 'identity'), (20, 256, 'random'), (10, 192, 'random')] + ([(40, 384, 'identity')] if '--big' in sys.argv else [])
    for blocks, ch, bn in nets:
        w = nsg.weights.make_random(blocks, ch, seed=0, bn=bn)
        bb = nsg.positions.game_positions(2, seed=5)
        planes = oracle_lib.load().extract_bits(bb)
        ref, _ = forward(w, planes, 'f64')
        for mode in ('f16', 'f16x3', 'f16m8', 'f16m6'):
            out, amax = forward(w, planes, mode)
            print(json.dumps({'net': f'{blocks}x{ch}', 'bn': bn, 'mode': mode, 'policy_max_abs_err': float(np.abs(out - ref).max()),
                              'policy_rms_err': float(np.sqrt(((out - ref) ** 2).mean())), 'logit_range': float(np.abs(ref).max()),
                              'act_max': amax}), flush=True)
