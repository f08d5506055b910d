"""TEST INFRASTRUCTURE — layer-local restatement of the ESRGAN hot path (forward AND backward), used to hold
every device kernel to the rounding it is entitled to at the BASELINE shapes.

NOT product code (same rule as oracle/esrgan_oracle.py: only tests/, smoke() and bench.py's checker legs import it).

Why layer-local.  An end-to-end comparison of a 351-conv bf16 network can only be held to ~1e-2: a bf16 store
that lands on the other side of a rounding boundary (because fp32 partial sums were added in another order) is a
1-ulp = 2^-8 difference that the following layers amplify.  Here every layer is recomputed on the CPU from the
*device's own input buffers* of that layer (exact bf16 / fp32 values copied back), so the only legitimate
difference is the accumulation order inside ONE layer: <= 1 bf16 ulp of the stored value in bf16 mode,
~1e-6 relative in the fp32 modes.  A wrong tap, a wrong residual weight, a missing mask or a mis-rounded
operand shows up as a percent-level error in exactly the layer that has it.

The formulas restate, operation by operation:
  forward   /root/reference/ssr/archs/rrdbnet_arch.py:37-44 (ResidualDenseBlock), :63-68 (RRDB), :116-137 (SSR_RRDBNet),
            /root/reference/ssr/archs/discriminator_arch.py:42-71 (SSR_UNetDiscriminatorSN)
  backward  what torch.autograd derives for those forwards (conv2d_input / conv2d_weight of every conv, LeakyReLU'
            = 1 | 0.2 by the sign of the output, nearest/bilinear interpolation adjoints, fan-in sums), with the
            storage points of the HIP path: every buffer is rounded once, when it is written.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

SLOPE = 0.2


def rnd(x: torch.Tensor, mode: str) -> torch.Tensor:
    """Storage rounding of the mode: bf16 buffers in 'bf16', fp32 buffers otherwise."""
    return x.to(torch.bfloat16).to(torch.float32) if mode == "bf16" else x


def lrelu(x):
    return F.leaky_relu(x, SLOPE)


def mask_of(out: torch.Tensor) -> torch.Tensor:
    """LeakyReLU' recovered from the stored OUTPUT (sign-preserving for slope > 0; at 0: slope, as torch)."""
    return torch.where(out > 0, torch.ones_like(out), torch.full_like(out, SLOPE))


def up2(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")


def sum2x2(g):
    """adjoint of nearest x2."""
    n, c, h, w = g.shape
    return g.view(n, c, h // 2, 2, w // 2, 2).sum(dim=(3, 5))


def bil(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def bil_T(g):
    """adjoint of bilinear x2 (align_corners=False), by autograd on the linear map."""
    n, c, h, w = g.shape
    z = torch.zeros(n, c, h // 2, w // 2, requires_grad=True)
    (gz,) = torch.autograd.grad(bil(z), z, g)
    return gz


def conv(x, w, b=None, stride=1, pad=1):
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def conv_T(w, dy, in_hw, stride=1, pad=1):
    """dL/dx of conv(x, w) given dL/dy."""
    n = dy.shape[0]
    return conv2d_input((n, w.shape[1], in_hw[0], in_hw[1]), w, dy, stride=stride, padding=pad)


def conv_dw(x, dy, wshape, stride=1, pad=1):
    return conv2d_weight(x, wshape, dy, stride=stride, padding=pad)


def bf16_ulp(v: torch.Tensor) -> torch.Tensor:
    """spacing of bfloat16 (8 significant bits) at |v|"""
    return torch.exp2(torch.floor(torch.log2(v.abs().clamp_min(1e-38))) - 7)


class Report:
    """Collects per-layer errors.  Row = (layer, max|dev - ref| / max|ref|, mean|dev - ref| / max|ref|, fraction of elements
    that differ at all, largest difference in bf16 ulps of the element — ignoring differences below 1e-5 * max|ref|, where
    fp32 cancellation noise of a near-zero sum can exceed the element's own ulp)."""

    def __init__(self):
        self.rows: List[tuple] = []

    def add(self, what: str, dev: torch.Tensor, ref: torch.Tensor):
        dev, ref = dev.float(), ref.float()
        assert dev.shape == ref.shape, (what, dev.shape, ref.shape)
        scale = float(ref.abs().max()) + 1e-30
        d = (dev - ref).abs()
        big = d > 1e-5 * scale
        ulps = float((d[big] / bf16_ulp(torch.maximum(ref.abs(), dev.abs())[big])).max()) if bool(big.any()) else 0.0
        self.rows.append((what, float(d.max()) / scale, float(d.mean()) / scale, float((d > 0).float().mean()), ulps))

    def check_bf16(self, max_ulps: float = 1.0, max_frac: float = 2e-3, tol_mean: float = 1e-5):
        """bf16 storage: a stored value may land on the adjacent bf16 value (the fp32 sum it rounds was accumulated in another
        order) — never further, only on a small fraction of the elements, and without any systematic component."""
        bad = [r for r in self.rows if r[4] > max_ulps or r[3] > max_frac or r[2] > tol_mean]
        assert not bad, f"{len(bad)} of {len(self.rows)} layers out of tolerance (ulps {max_ulps}, differing fraction {max_frac}, " \
                        f"mean {tol_mean}); worst: {sorted(bad, key=lambda r: -r[4])[:6]}"

    def worst(self, prefix: str = ""):
        rows = [r for r in self.rows if r[0].startswith(prefix)]
        return max(rows, key=lambda r: r[1]) if rows else None

    def check(self, tol_max: float, tol_mean: float, prefix: str = ""):
        bad = [r for r in self.rows if r[0].startswith(prefix) and (r[1] > tol_max or r[2] > tol_mean)]
        assert not bad, f"{len(bad)} of {len(self.rows)} layers out of tolerance (max {tol_max}, mean {tol_mean}); worst: " \
                        f"{sorted(bad, key=lambda r: -r[1])[:6]}"

    def summary(self) -> Dict[str, float]:
        return {"layers": len(self.rows), "worst_max": max(r[1] for r in self.rows),
                "worst_mean": max(r[2] for r in self.rows), "worst_frac_differing": max(r[3] for r in self.rows),
                "worst_ulps": max(r[4] for r in self.rows), "worst_layer": max(self.rows, key=lambda r: r[1])[0]}


# ---------------------------------------------------------------------------------------------------------
# Generator
# ---------------------------------------------------------------------------------------------------------
def generator_forward_layers(sd, bufs: Dict[str, torch.Tensor], nf: int, gc: int, nb: int, mode: str, rep: Report):
    """`bufs`: the device's NCHW copies: xin, rdb[r] (nf+4gc channels, r = 0..3nb-1), body_out, trunk, ups[i], hr, out.
    Every stored tensor is recomputed from the device's inputs of its layer."""
    w = lambda name: rnd(sd[name + ".weight"], mode)
    b = lambda name: sd[name + ".bias"]
    rep.add("fwd conv_first", bufs["rdb"][0][:, :nf], rnd(conv(bufs["xin"], w("conv_first"), b("conv_first")), mode))
    n_rdb = 3 * nb
    for r in range(n_rdb):
        i, j = divmod(r, 3)
        p = f"body.{i}.rdb{j + 1}"
        cur = bufs["rdb"][r]
        for k in range(1, 5):   # x_k = lrelu(conv_k(cat(x, x1..x_{k-1})))                       rrdbnet_arch.py:39-42
            cin = nf + (k - 1) * gc
            ref = rnd(lrelu(conv(cur[:, :cin], w(f"{p}.conv{k}"), b(f"{p}.conv{k}"))), mode)
            rep.add(f"fwd {p}.conv{k}", cur[:, cin:cin + gc], ref)
        x5 = conv(cur, w(f"{p}.conv5"), b(f"{p}.conv5"))
        if j < 2:               # x5 * 0.2 + x                                                    :44
            ref = x5 * 0.2 + cur[:, :nf]
        else:                   # (x5 * 0.2 + x) * 0.2 + x_rrdb, one fused epilogue               :44, :68
            ref = x5 * 0.04 + cur[:, :nf] * 0.2 + bufs["rdb"][r - 2][:, :nf]
        dst = bufs["body_out"] if r == n_rdb - 1 else bufs["rdb"][r + 1][:, :nf]
        rep.add(f"fwd {p}.conv5", dst, rnd(ref, mode))
    feat = bufs["rdb"][0][:, :nf]
    rep.add("fwd conv_body", bufs["trunk"], rnd(feat + conv(bufs["body_out"], w("conv_body"), b("conv_body")), mode))   # :125
    src = bufs["trunk"]
    for i, u in enumerate(bufs["ups"]):   # lrelu(conv(nearest x2))                              :127-128
        nm = f"conv_up{i + 1}"
        rep.add(f"fwd {nm}", u, rnd(lrelu(conv(up2(src), w(nm), b(nm))), mode))
        src = u
    rep.add("fwd conv_hr", bufs["hr"], rnd(lrelu(conv(src, w("conv_hr"), b("conv_hr"))), mode))
    rep.add("fwd conv_last", bufs["out"], rnd(conv(bufs["hr"], w("conv_last"), b("conv_last")), mode))


def generator_backward_layers(sd, bufs, gb: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], nf, gc, nb, mode,
                              rep: Report, fused_bwd_weights: bool = True):
    """`gb`: the device's gradient buffers: d_out, g_hr, g_ups[i], g_tmp[i], g_trunk, g_body_out, drdb[r] (pre-activation
    gradients of x1..x4 in channels nf.., gradient of the block input in channels 0..nf).  `grads`: the device's
    parameter gradients (reference key layout)."""
    w = lambda name: rnd(sd[name + ".weight"], mode)
    hw = lambda t: t.shape[-2:]
    n_up = len(bufs["ups"])
    last_up = bufs["ups"][-1]
    # conv_last / conv_hr
    rep.add("bwd conv_last dgrad", gb["g_hr"], rnd(mask_of(bufs["hr"]) * conv_T(w("conv_last"), gb["d_out"], hw(bufs["hr"])), mode))
    rep.add("bwd conv_hr dgrad", gb["g_ups"][-1], rnd(mask_of(last_up) * conv_T(w("conv_hr"), gb["g_hr"], hw(last_up)), mode))
    rep.add("wgrad conv_last", grads["conv_last.weight"], conv_dw(bufs["hr"], gb["d_out"], sd["conv_last.weight"].shape))
    rep.add("bgrad conv_last", grads["conv_last.bias"], gb["d_out"].sum(dim=(0, 2, 3)))
    rep.add("wgrad conv_hr", grads["conv_hr.weight"], conv_dw(last_up, gb["g_hr"], sd["conv_hr.weight"].shape))
    rep.add("bgrad conv_hr", grads["conv_hr.bias"], gb["g_hr"].sum(dim=(0, 2, 3)))
    for i in reversed(range(n_up)):
        nm = f"conv_up{i + 1}"
        src = bufs["ups"][i - 1] if i > 0 else bufs["trunk"]
        rep.add(f"bwd {nm} dgrad", gb["g_tmp"][i], rnd(conv_T(w(nm), gb["g_ups"][i], hw(gb["g_ups"][i])), mode))
        s = sum2x2(gb["g_tmp"][i])
        if i > 0:
            rep.add(f"bwd {nm} nearest^T", gb["g_ups"][i - 1], rnd(mask_of(src) * s, mode))
        else:
            rep.add(f"bwd {nm} nearest^T", gb["g_trunk"], rnd(s, mode))
        rep.add(f"wgrad {nm}", grads[nm + ".weight"], conv_dw(up2(src), gb["g_ups"][i], sd[nm + ".weight"].shape))
        rep.add(f"bgrad {nm}", grads[nm + ".bias"], gb["g_ups"][i].sum(dim=(0, 2, 3)))
    rep.add("bwd conv_body dgrad", gb["g_body_out"], rnd(conv_T(w("conv_body"), gb["g_trunk"], hw(bufs["body_out"])), mode))
    rep.add("wgrad conv_body", grads["conv_body.weight"], conv_dw(bufs["body_out"], gb["g_trunk"], sd["conv_body.weight"].shape))
    n_rdb = 3 * nb
    H, W = hw(bufs["body_out"])
    for r in reversed(range(n_rdb)):
        i, j = divmod(r, 3)
        p = f"body.{i}.rdb{j + 1}"
        cur, dcur = bufs["rdb"][r], gb["drdb"][r]
        d_out = gb["g_body_out"] if r == n_rdb - 1 else gb["drdb"][r + 1][:, :nf]
        a5, b5 = (0.04, 0.2) if j == 2 else (0.2, 1.0)
        rr = 3 * i + 2
        d_rrdb = gb["g_body_out"] if rr == n_rdb - 1 else gb["drdb"][rr + 1][:, :nf]
        # conv5's scale is folded into the weights the backward kernel reads (engine.ParamStore.add_rdb_gather):
        # round(a5 * W5) in bf16 mode
        w5 = rnd(sd[f"{p}.conv5.weight"] * a5, mode) if fused_bwd_weights else w(f"{p}.conv5") * a5
        gin = {5: conv_T(w5, d_out, (H, W))}                 # gradient w.r.t. conv5's 192-channel input
        for k in (4, 3, 2, 1):                               # dpre_k = lrelu'(x_k) * sum_{j>k} conv_j^T[dpre_j]
            c0 = nf + (k - 1) * gc
            tot = sum(gin[jj][:, c0:c0 + gc] for jj in gin)
            rep.add(f"bwd {p} dpre{k}", dcur[:, c0:c0 + gc], rnd(mask_of(cur[:, c0:c0 + gc]) * tot, mode))
            # the next contribution uses the DEVICE's dpre_k (layer-local)
            gin[k] = conv_T(w(f"{p}.conv{k}"), dcur[:, c0:c0 + gc], (H, W))
        tot = sum(gin[jj][:, :nf] for jj in gin) + b5 * d_out      # `+ x` path of the block           :44
        if j == 0:
            tot = tot + d_rrdb                                      # `+ x` path of the RRDB            :68
        ref = rnd(tot, mode)
        if r == 0:                                                  # d feat += d trunk                 :125
            ref = rnd(ref + gb["g_trunk"], mode)
        rep.add(f"bwd {p} dx", dcur[:, :nf], ref)
        # weight gradients: x = dense buffer prefix, dy = pre-activation gradient (conv5: a5 * d_out)
        for k in (1, 2, 3, 4):
            cin = nf + (k - 1) * gc
            rep.add(f"wgrad {p}.conv{k}", grads[f"{p}.conv{k}.weight"],
                    conv_dw(cur[:, :cin], dcur[:, cin:cin + gc], sd[f"{p}.conv{k}.weight"].shape))
        rep.add(f"wgrad {p}.conv5", grads[f"{p}.conv5.weight"], a5 * conv_dw(cur, d_out, sd[f"{p}.conv5.weight"].shape))
        rep.add(f"bgrad {p}.conv5", grads[f"{p}.conv5.bias"], a5 * d_out.sum(dim=(0, 2, 3)))
    rep.add("wgrad conv_first", grads["conv_first.weight"],
            conv_dw(bufs["xin"], gb["drdb"][0][:, :nf], sd["conv_first.weight"].shape))


# ---------------------------------------------------------------------------------------------------------
# Discriminator
# ---------------------------------------------------------------------------------------------------------
D_STRIDES = {"conv1": 2, "conv2": 2, "conv3": 2}


def discriminator_forward_layers(wts: Dict[str, torch.Tensor], bias: Dict[str, torch.Tensor], x_in, bufs, skip: bool,
                                 mode: str, rep: Report):
    """`wts[name]`: the weight the conv sees BEFORE packing (W, or W_orig / sigma for the spectral-normalised layers, with
    the device's sigma); `bufs`: x0, x1, x2, x3, u3, a4, u4, a5, u5, a6, x6, o7, o8, logits (NCHW)."""
    w = lambda n: rnd(wts[n], mode)
    rep.add("fwd D.conv0", bufs["x0"], rnd(lrelu(conv(x_in, w("conv0"), bias["conv0"])), mode))
    rep.add("fwd D.conv1", bufs["x1"], rnd(lrelu(conv(bufs["x0"], w("conv1"), None, 2)), mode))
    rep.add("fwd D.conv2", bufs["x2"], rnd(lrelu(conv(bufs["x1"], w("conv2"), None, 2)), mode))
    rep.add("fwd D.conv3", bufs["x3"], rnd(lrelu(conv(bufs["x2"], w("conv3"), None, 2)), mode))
    rep.add("fwd D.bilinear3", bufs["u3"], rnd(bil(bufs["x3"]), mode))                       # discriminator_arch.py:50
    rep.add("fwd D.conv4", bufs["a4"], rnd(lrelu(conv(bufs["u3"], w("conv4"))), mode))
    rep.add("fwd D.bilinear4", bufs["u4"], rnd(bil(bufs["a4"] + bufs["x2"] if skip else bufs["a4"]), mode))   # :53-55
    rep.add("fwd D.conv5", bufs["a5"], rnd(lrelu(conv(bufs["u4"], w("conv5"))), mode))
    rep.add("fwd D.bilinear5", bufs["u5"], rnd(bil(bufs["a5"] + bufs["x1"] if skip else bufs["a5"]), mode))   # :57-60
    pre6 = lrelu(conv(bufs["u5"], w("conv6")))
    rep.add("fwd D.conv6", bufs["a6"], rnd(pre6, mode))
    if skip:
        rep.add("fwd D.conv6+skip", bufs["x6"], rnd(pre6 + bufs["x0"], mode))                 # :62-64
    x6 = bufs["x6"] if skip else bufs["a6"]
    rep.add("fwd D.conv7", bufs["o7"], rnd(lrelu(conv(x6, w("conv7"))), mode))
    rep.add("fwd D.conv8", bufs["o8"], rnd(lrelu(conv(bufs["o7"], w("conv8"))), mode))
    rep.add("fwd D.conv9", bufs["logits"], rnd(conv(bufs["o8"], w("conv9"), bias["conv9"]), mode))


def discriminator_backward_layers(wts, x_in, bufs, gb, wgrads: Optional[Dict[str, torch.Tensor]], skip: bool, mode: str,
                                  rep: Report, in_residual: Optional[torch.Tensor] = None):
    """`gb`: d_logits, g_o8, g_o7, g_a6, g_x6, g_u5, g_a5, g_x5, g_u4, g_a4, g_x4, g_u3, g3, g2, g1, g0, g_in (NCHW).
    `wgrads[name]`: device gradient w.r.t. the weight the conv sees (normalised weight for SN layers), or None when the
    parameters are frozen (generator phase, ssr_esrgan_model.py:136-137)."""
    w = lambda n: rnd(wts[n], mode)
    hw = lambda t: t.shape[-2:]
    rep.add("bwd D.conv9 dgrad", gb["g_o8"], rnd(mask_of(bufs["o8"]) * conv_T(w("conv9"), gb["d_logits"], hw(bufs["o8"])), mode))
    rep.add("bwd D.conv8 dgrad", gb["g_o7"], rnd(mask_of(bufs["o7"]) * conv_T(w("conv8"), gb["g_o8"], hw(bufs["o7"])), mode))
    s = conv_T(w("conv7"), gb["g_o7"], hw(bufs["a6"]))
    rep.add("bwd D.conv7 dgrad", gb["g_a6"], rnd(mask_of(bufs["a6"]) * s, mode))
    if skip:
        rep.add("bwd D.conv7 dgrad(skip)", gb["g_x6"], rnd(s, mode))
    rep.add("bwd D.conv6 dgrad", gb["g_u5"], rnd(conv_T(w("conv6"), gb["g_a6"], hw(bufs["u5"])), mode))
    t = bil_T(gb["g_u5"])
    rep.add("bwd D.bilinear5", gb["g_a5"], rnd(mask_of(bufs["a5"]) * t, mode))
    if skip:
        rep.add("bwd D.bilinear5(skip)", gb["g_x5"], rnd(t, mode))
    rep.add("bwd D.conv5 dgrad", gb["g_u4"], rnd(conv_T(w("conv5"), gb["g_a5"], hw(bufs["u4"])), mode))
    t = bil_T(gb["g_u4"])
    rep.add("bwd D.bilinear4", gb["g_a4"], rnd(mask_of(bufs["a4"]) * t, mode))
    if skip:
        rep.add("bwd D.bilinear4(skip)", gb["g_x4"], rnd(t, mode))
    rep.add("bwd D.conv4 dgrad", gb["g_u3"], rnd(conv_T(w("conv4"), gb["g_a4"], hw(bufs["u3"])), mode))
    rep.add("bwd D.bilinear3", gb["g3"], rnd(mask_of(bufs["x3"]) * bil_T(gb["g_u3"]), mode))
    z = lambda k: gb[k] if skip else 0.0
    rep.add("bwd D.conv3 dgrad", gb["g2"], rnd(mask_of(bufs["x2"]) * (conv_T(w("conv3"), gb["g3"], hw(bufs["x2"]), 2) + z("g_x4")), mode))
    rep.add("bwd D.conv2 dgrad", gb["g1"], rnd(mask_of(bufs["x1"]) * (conv_T(w("conv2"), gb["g2"], hw(bufs["x1"]), 2) + z("g_x5")), mode))
    rep.add("bwd D.conv1 dgrad", gb["g0"], rnd(mask_of(bufs["x0"]) * (conv_T(w("conv1"), gb["g1"], hw(bufs["x0"]), 2) + z("g_x6")), mode))
    if "g_in" in gb:
        ref = conv_T(w("conv0"), gb["g0"], hw(x_in))
        if in_residual is not None:
            ref = ref + in_residual
        rep.add("bwd D.conv0 dgrad", gb["g_in"], rnd(ref, mode))
    if wgrads is None:
        return
    x6 = bufs["x6"] if skip else bufs["a6"]
    pairs = [("conv9", bufs["o8"], gb["d_logits"], 1), ("conv8", bufs["o7"], gb["g_o8"], 1), ("conv7", x6, gb["g_o7"], 1),
             ("conv6", bufs["u5"], gb["g_a6"], 1), ("conv5", bufs["u4"], gb["g_a5"], 1), ("conv4", bufs["u3"], gb["g_a4"], 1),
             ("conv3", bufs["x2"], gb["g3"], 2), ("conv2", bufs["x1"], gb["g2"], 2), ("conv1", bufs["x0"], gb["g1"], 2),
             ("conv0", x_in, gb["g0"], 1)]
    for name, x, dy, stride in pairs:
        rep.add(f"wgrad D.{name}", wgrads[name], conv_dw(x, dy, wts[name].shape, stride))
