"""CPU: pins oracle/layerwise.py (the layer-local forward/backward restatement the GPU parity tests use at the
BASELINE shapes) against torch.autograd over oracle/esrgan_oracle.py, which tests/test_oracle_golden.py pins against
the golden vectors of the unmodified reference classes.

A recording precision model captures every tensor the oracle "stores" (and, after backward(), the gradient that
arrives there); those play the role of the device buffers.  layerwise must then reproduce every one of them from
its neighbours: exactly the check the GPU tests run with real device buffers."""
from collections import OrderedDict

import pytest
import torch
import torch.nn.functional as F

from oracle import esrgan_oracle as O
from oracle import layerwise as LW


class Rec(O.Prec):
    """Wraps a precision model and records (stored value, tensor whose .grad is the buffer's gradient)."""

    def __init__(self, base):
        self.base, self.name, self.log = base, base.name, []

    def a(self, x):
        y = self.base.a(x)
        if y.requires_grad:
            x.retain_grad()
        self.log.append((y, x))        # gradient buffer = gradient after the backward rounding = grad of the INPUT of the hook
        return y

    def act(self, pre):
        if pre.requires_grad:
            pre.retain_grad()
        y = self.base.act(pre)
        self.log.append((y, pre))      # buffers hold pre-activation gradients
        return y

    def w(self, x):
        return self.base.w(x)

    def g(self, x):
        return self.base.g(x)


def _grad(t):
    return t.grad if t.grad is not None else torch.zeros_like(t)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_generator_layerwise_matches_autograd(mode):
    nf, gc, nb = 16, 8, 2
    kw = dict(num_in_ch=5, num_out_ch=3, scale=4, num_feat=nf, num_block=nb, num_grow_ch=gc)
    sd = O.generator_init(seed=3, **kw)
    for k in list(sd):
        if k.endswith(".bias"):
            sd[k] = torch.randn_like(sd[k]) * 0.05
    sdg = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    torch.manual_seed(0)
    x = torch.rand(2, 5, 8, 12).requires_grad_(True)
    rec = Rec(O.BF16 if mode == "bf16" else O.FP32)
    y = O.generator_forward(sdg, x, 4, rec)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    log = rec.log
    # order of the hooks in generator_forward: xin, conv_first, per RRDB: [4 acts, out] x2, 4 acts, rrdb out; conv_body(trunk),
    # up1, up2, hr, out
    it = iter(log)
    xin = next(it)
    feat = next(it)
    blocks, outs = [], []
    for i in range(nb):
        for j in range(3):
            blocks.append([next(it) for _ in range(4)])
            outs.append(next(it))
    trunk, up1, up2_, hr, out = next(it), next(it), next(it), next(it), next(it)
    n_rdb = 3 * nb
    inputs = [feat] + outs[:-1]
    val = lambda p: p[0].detach()
    bufs = {"xin": val(xin), "rdb": [torch.cat([val(inputs[r])] + [val(b) for b in blocks[r]], 1) for r in range(n_rdb)],
            "body_out": val(outs[-1]), "trunk": val(trunk), "ups": [val(up1), val(up2_)], "hr": val(hr), "out": val(out)}
    rep = LW.Report()
    LW.generator_forward_layers(sd, bufs, nf, gc, nb, mode, rep)
    tol = 1e-5 if mode == "fp32" else 4e-3
    rep.check(tol, tol)
    # gradient buffers.  The block-input gradient of RDB r (channels 0..nf of drdb[r]) is the gradient arriving at the
    # stored tensor inputs[r]
    g = lambda p: _grad(p[1]).detach()
    gb = {"d_out": g(out), "g_hr": g(hr), "g_ups": [g(up1), g(up2_)], "g_trunk": g(trunk), "g_body_out": g(outs[-1]),
          "drdb": [torch.cat([g(inputs[r])] + [g(b) for b in blocks[r]], 1) for r in range(n_rdb)]}
    # g_tmp (the dgrad of an upsampling conv on the fine grid) has no autograd twin: take layerwise's own value and check the
    # two-stage result instead
    w = lambda name: LW.rnd(sd[name + ".weight"], mode)
    gb["g_tmp"] = [LW.rnd(LW.conv_T(w("conv_up1"), gb["g_ups"][0], gb["g_ups"][0].shape[-2:]), mode),
                   LW.rnd(LW.conv_T(w("conv_up2"), gb["g_ups"][1], gb["g_ups"][1].shape[-2:]), mode)]
    grads = {k: v.grad.detach() for k, v in sdg.items()}
    rep = LW.Report()
    LW.generator_backward_layers(sd, bufs, gb, grads, nf, gc, nb, mode, rep, fused_bwd_weights=False)
    # bf16: the device stores the fine-grid dgrad of an upsampling conv (g_tmp) before the 2x2 sum, autograd rounds the
    # summed gradient once: those two rows legitimately differ by one bf16 ulp (2^-7 relative at worst)
    two_stage = [r for r in rep.rows if r[0].endswith("nearest^T")]
    rep.rows = [r for r in rep.rows if not r[0].endswith("nearest^T")]
    rep.check(tol, tol)
    assert len(two_stage) == 2 and all(r[1] <= (1e-5 if mode == "fp32" else 2 ** -7) for r in two_stage), two_stage
    assert len(rep.rows) > 10 * n_rdb


@pytest.mark.parametrize("skip", [True, False])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_discriminator_layerwise_matches_autograd(mode, skip):
    nf = 8
    sd = O.discriminator_init(6, nf, seed=5)
    sdg = OrderedDict((k, (v.clone().requires_grad_(True) if k in O.D_PARAM_KEYS else v.clone())) for k, v in sd.items())
    torch.manual_seed(1)
    x = torch.rand(2, 6, 16, 24).requires_grad_(True)
    rec = Rec(O.BF16 if mode == "bf16" else O.FP32)
    # normalised weights as the forward sees them (same power iteration: update_buffers=False keeps sd's u, v)
    wts, wts_leaf = {}, {}
    for n in O.SN_LAYERS:
        w_sn, _, _ = O.spectral_norm_weight(sd[n + ".weight_orig"], sd[n + ".weight_u"], sd[n + ".weight_v"], True)
        wts[n] = w_sn
    wts["conv0"], wts["conv9"] = sd["conv0.weight"], sd["conv9.weight"]
    # run the oracle forward on leaf copies of the normalised weights so that .grad is the gradient w.r.t. them
    leaf = {n: wts[n].clone().requires_grad_(True) for n in wts}
    y = _d_forward_with_weights(leaf, {"conv0": sd["conv0.bias"], "conv9": sd["conv9.bias"]}, x, skip, rec)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    names = ["xin", "x0", "x1", "x2", "x3", "u3", "a4", "u4", "a5", "u5"] + (["x6"] if skip else ["a6"]) + ["o7", "o8", "logits"]
    assert len(rec.log) == len(names)
    L = dict(zip(names, rec.log))
    val = lambda k: L[k][0].detach()
    g = lambda k: _grad(L[k][1]).detach()
    bufs = {k: val(k) for k in names if k != "xin"}
    if skip:   # a6 is stored separately by the device (the backward mask); recompute it from its definition
        bufs["a6"] = LW.rnd(LW.lrelu(LW.conv(bufs["u5"], LW.rnd(wts["conv6"], mode))), mode)
    else:
        bufs["x6"] = bufs["a6"]
    rep = LW.Report()
    LW.discriminator_forward_layers(wts, {"conv0": sd["conv0.bias"], "conv9": sd["conv9.bias"]}, val("xin"), bufs, skip, mode, rep)
    tol = 1e-5 if mode == "fp32" else 4e-3
    rep.check(tol, tol)
    gb = {"d_logits": g("logits"), "g_o8": g("o8"), "g_o7": g("o7"), "g_u5": g("u5"), "g_a5": g("a5"), "g_u4": g("u4"),
          "g_a4": g("a4"), "g_u3": g("u3"), "g3": g("x3"), "g2": g("x2"), "g1": g("x1"), "g0": g("x0"), "g_in": g("xin")}
    # buffers without an autograd twin are taken from layerwise's own definition (their consumers are then checked)
    s = LW.conv_T(LW.rnd(wts["conv7"], mode), gb["g_o7"], bufs["a6"].shape[-2:])
    gb["g_a6"] = LW.rnd(LW.mask_of(bufs["a6"]) * s, mode)
    if skip:
        gb["g_x6"] = LW.rnd(s, mode)
        gb["g_x5"] = LW.rnd(LW.bil_T(gb["g_u5"]), mode)
        gb["g_x4"] = LW.rnd(LW.bil_T(gb["g_u4"]), mode)
    wgr = {n: leaf[n].grad.detach() for n in leaf}
    rep = LW.Report()
    LW.discriminator_backward_layers(wts, val("xin"), bufs, gb, wgr, skip, mode, rep)
    rep.check(tol, tol)


def _d_forward_with_weights(w, bias, x, skip, prec):
    """discriminator_forward of the oracle with the (already normalised) weights given explicitly — same statements as
    oracle/esrgan_oracle.py:discriminator_forward after the spectral-norm loop."""
    a = prec.a
    sn = lambda name, t, stride: F.conv2d(t, prec.w(w[name]), None, stride=stride, padding=1)
    bil = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    x0 = prec.act(F.conv2d(a(x), prec.w(w["conv0"]), bias["conv0"], padding=1))
    x1 = prec.act(sn("conv1", x0, 2))
    x2 = prec.act(sn("conv2", x1, 2))
    x3 = prec.act(sn("conv3", x2, 2))
    x3 = a(bil(x3))
    x4 = prec.act(sn("conv4", x3, 1))
    if skip:
        x4 = x4 + prec.g(x2)
    x4 = a(bil(x4))
    x5 = prec.act(sn("conv5", x4, 1))
    if skip:
        x5 = x5 + prec.g(x1)
    x5 = a(bil(x5))
    if skip:
        x6 = prec.w(O._lrelu(prec.g(sn("conv6", x5, 1))) + prec.g(x0))
        prec.log.append((x6, x6))
    else:
        x6 = prec.act(sn("conv6", x5, 1))
    out = prec.act(sn("conv7", x6, 1))
    out = prec.act(sn("conv8", out, 1))
    return a(F.conv2d(out, prec.w(w["conv9"]), bias["conv9"], padding=1))


def test_d_forward_helper_is_the_oracle_forward():
    """the helper above must be the oracle's discriminator_forward (same result bit for bit)."""
    sd = O.discriminator_init(3, 8, seed=2)
    x = torch.rand(1, 3, 16, 16)
    wts = {}
    for n in O.SN_LAYERS:
        wts[n], _, _ = O.spectral_norm_weight(sd[n + ".weight_orig"], sd[n + ".weight_u"], sd[n + ".weight_v"], True)
    wts["conv0"], wts["conv9"] = sd["conv0.weight"], sd["conv9.weight"]
    bias = {"conv0": sd["conv0.bias"], "conv9": sd["conv9.bias"]}
    for prec in (O.FP32, O.BF16):
        for skip in (True, False):
            rec = Rec(prec)
            a = _d_forward_with_weights(wts, bias, x, skip, rec)
            b = O.discriminator_forward(dict(sd), x, train=True, skip_connection=skip, update_buffers=False, prec=prec)
            assert torch.equal(a, b)
