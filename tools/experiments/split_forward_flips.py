#!/usr/bin/env python3
"""CPU experiment behind the forward arithmetic of the all-gates mode: which split of the forward products keeps the parameter
gradients inside the reference's 1e-3 gate?  The gradients leave the gate through flipped LeakyReLU decisions only,
so for every candidate arithmetic the generator forward is EMULATED on the CPU (operand splits as the matrix cores would see them, fp32
accumulation), its decisions are recorded, and the float64 oracle is differentiated WITH those decisions (MaskedPrec) and compared with
the float64 truth that takes its own decisions.  Usage: split_forward_flips.py [B=2] [nb=23] [cin=24]"""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import esrgan_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 23
CIN = int(sys.argv[3]) if len(sys.argv) > 3 else 24
torch.manual_seed(0)
kw = dict(num_in_ch=CIN, num_out_ch=3, scale=4, num_feat=64, num_block=NB, num_grow_ch=32)
sd = O.generator_init(seed=0, **kw)
x = torch.rand(B, CIN, 32, 32)
g = torch.randn(B, 3, 128, 128)


FTZ = False        # emulate a matrix core that flushes fp16 subnormal inputs (|p| < 2^-14) to zero


def split(t, dt, n):
    parts, r = [], t
    for _ in range(n):
        p = r.to(dt).float()
        if FTZ and dt == torch.float16:
            p = torch.where(p.abs() < 2.0 ** -14, torch.zeros_like(p), p)
        parts.append(p)
        r = r - p
    return parts


def make_conv(kind):
    """kind: (dtype, pieces, products kept as (i, j) with i + j <= order, weight scale exponent, activation scale exponent)"""
    if kind is None:
        return None
    dt, n, order, kw_, ka_ = kind[:5]
    ftz = len(kind) > 5 and kind[5]

    def conv(sd_, name, xx, stride=1, pad=1, prec=None):
        global FTZ
        FTZ = ftz
        w = sd_[name + ".weight"]
        xs = split(xx * (2.0 ** ka_), dt, n)
        ws = split(w * (2.0 ** kw_), dt, n)
        acc = None
        terms = sorted(((i, j) for i in range(n) for j in range(n) if i + j <= order), key=lambda p: -(p[0] + p[1]))
        for i, j in terms:                      # small terms first, fp32 accumulation
            t = F.conv2d(xs[i], ws[j], None, stride=stride, padding=pad)
            acc = t if acc is None else acc + t
        acc = acc * (2.0 ** -(kw_ + ka_))
        b = sd_.get(name + ".bias")
        return acc if b is None else acc + b.view(1, -1, 1, 1)
    return conv


class Rec(O.Prec):
    def __init__(self):
        self.masks = []

    def act(self, pre):
        self.masks.append(pre.detach() > 0)
        return F.leaky_relu(pre, O.LRELU_SLOPE)
    lrelu_raw = act


def grads64(prec):
    sdg = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    (O.generator_forward(sdg, x.double(), 4, prec) * g.double()).sum().backward()
    return {k: v.grad for k, v in sdg.items()}


t0 = time.time()
truth_rec = Rec()
sdg = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
(O.generator_forward(sdg, x.double(), 4, truth_rec) * g.double()).sum().backward()
truth = {k: v.grad for k, v in sdg.items()}
nmask = sum(m.numel() for m in truth_rec.masks)
print(f"float64 truth: {time.time() - t0:.1f} s, {nmask} decisions, B={B} nb={NB} cin={CIN}", flush=True)

cands = [
    ("fp32 (plain CPU conv: 'an fp32 evaluation')", None),
    ("bf16 x3 (hi, lo; fp32x3 today)", (torch.bfloat16, 2, 1, 0, 0)),
    ("bf16 x6 (hi, mid, lo; order <= 2)", (torch.bfloat16, 3, 2, 0, 0)),
    ("fp16 x3 unscaled", (torch.float16, 2, 1, 0, 0)),
    ("fp16 x3, weights x 2^10", (torch.float16, 2, 1, 10, 0)),
    ("fp16 x3, weights x 2^10, activations x 2^3", (torch.float16, 2, 1, 10, 3)),
    ("fp16 x4 (+ lo.lo), weights x 2^10", (torch.float16, 2, 2, 10, 0)),
    ("FTZ fp16 x3, weights x 2^10", (torch.float16, 2, 1, 10, 0, True)),
    ("FTZ fp16 x3, weights x 2^10, activations x 2^3", (torch.float16, 2, 1, 10, 3, True)),
    ("FTZ fp16 x3, weights x 2^10, activations x 2^6", (torch.float16, 2, 1, 10, 6, True)),
    ("fp16 x3, weights x 2^6", (torch.float16, 2, 1, 6, 0)),
    ("fp16 x3, weights x 2^13", (torch.float16, 2, 1, 13, 0)),
]
if len(sys.argv) > 4:
    cands = [c for c in cands if any(t in c[0] for t in sys.argv[4].split(","))]
orig = O._conv
for name, kind in cands:
    t0 = time.time()
    rec = Rec()
    c = make_conv(kind)
    if c is not None:
        O._conv = c
    with torch.no_grad():
        y = O.generator_forward(sd, x, 4, rec)
    O._conv = orig
    flips = sum(int((a != b).sum()) for a, b in zip(rec.masks, truth_rec.masks))
    gm = grads64(O.MaskedPrec(rec.masks))
    worst = (0.0, 0.0, "")
    for k, r in truth.items():
        e = (gm[k] - r).abs()
        mx = float(r.abs().max())
        out = float((e > 1e-3 * mx + 1e-3 * r.abs()).double().mean())
        en = float(e.max() / mx)
        if (out, en) > worst[:2]:
            worst = (out, en, k)
    e1 = (gm["conv_first.weight"] - truth["conv_first.weight"]).abs()
    r1 = truth["conv_first.weight"]
    print(f"{name:52s} flips {flips:6d}  worst tensor: {100 * worst[0]:6.2f} % outside, max-norm {worst[1]:.2e} ({worst[2]});"
          f"  conv_first.weight {100 * float((e1 > 1e-3 * r1.abs().max() + 1e-3 * r1.abs()).double().mean()):6.2f} % / {float(e1.max() / r1.abs().max()):.2e}"
          f"   [{time.time() - t0:.0f} s]", flush=True)
