"""TEST INFRASTRUCTURE — CPU restatement (the oracle) of the reference's ESRGAN hot path.

NOT product code.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this module, and only as the checker / the reported CPU baseline.
The product path (satlas_super_resolution_amd) never imports anything under oracle/.

Everything here is plain fp32 PyTorch on the CPU, written functionally over a
reference-layout ``state_dict`` so that it needs neither BasicSR nor the reference
checkout (which does not exist on the GPU box).

Pinning status: the reference ships no tests, golden vectors or fixtures for this
path — its own tests leave parity unpinned.  This restatement is
therefore pinned against the *unmodified reference classes executed in the build
container* (oracle/ref_shim.py + oracle/make_golden.py -> tests/golden/*.pt, checked
by tests/test_oracle_golden.py).  The BasicSR pieces (GANLoss, L1Loss, model_ema,
Adam defaults) are restated from basicsr==1.4.2's published behaviour
(/root/reference/requirements.txt:1), which is not on disk: for those, parity is
anchored on the reference's call sites only (ssr_esrgan_model.py:147-231).

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2  # ssr/archs/rrdbnet_arch.py:32, discriminator_arch.py:44


# ----------------------------------------------------------------------------
# Precision model.  The reference computes in fp32 (`Prec()` = identity: the restatement proper).
# `BF16` models the throughput mode of the HIP path (BASELINE.json configs[1] "bf16"): the SAME
# algorithm with every tensor the device keeps in HBM rounded to bfloat16 at the point where it is
# stored (conv outputs after their fused epilogue, interpolation outputs, network inputs, the logits
# and the gradients flowing back through those same buffers), weights rounded where they are packed
# for the matrix core, and everything in between (accumulation, bias, LeakyReLU, residual sums,
# losses, spectral norm, Adam, EMA, master weights) in fp32.  It exists so that the bf16 kernels are
# held to the rounding they are entitled to (1 bf16 ulp per stored value) instead of a loose
# end-to-end tolerance against the fp32 result.
# ----------------------------------------------------------------------------
class _RoundBoth(torch.autograd.Function):
    """x -> bf16(x) going forward, g -> bf16(g) going back: a tensor AND its gradient live in bf16 buffers."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundFwd(torch.autograd.Function):
    """packed weights: rounded for the matrix core, gradient (an fp32 wgrad into the fp32 arena) untouched."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """identity forward, bf16 gradient: a gradient buffer without a forward twin (the L1 gradient)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class Prec:
    """fp32: all three hooks are the identity."""
    name = "fp32"

    def a(self, x):      # an activation stored in HBM (and its gradient)
        return x

    def w(self, x):      # a weight tensor as the matrix core sees it
        return x

    def g(self, x):      # a gradient-only buffer
        return x

    def act(self, pre):  # LeakyReLU output stored in HBM; its gradient buffer holds the PRE-activation gradient
        return F.leaky_relu(pre, LRELU_SLOPE)

    def act_relu(self, pre):  # same for the ReLU layers of the VGG19 feature extractor
        return F.relu(pre)

    def lrelu_raw(self, pre):  # a LeakyReLU whose output is not stored on its own (conv6 of the discriminator: lrelu(acc) + x0)
        return F.leaky_relu(pre, LRELU_SLOPE)


class _PrecBF16(Prec):
    name = "bf16"

    def a(self, x):
        return _RoundBoth.apply(x)

    def w(self, x):
        return _RoundFwd.apply(x)

    def g(self, x):
        return _RoundBwd.apply(x)

    def act(self, pre):
        # forward: round(lrelu(acc)) in the producer's epilogue; backward: the consumer-side epilogue applies the
        # LeakyReLU mask to the fp32 gradient sum and THEN rounds (the buffers hold pre-activation gradients)
        return _RoundFwd.apply(F.leaky_relu(_RoundBwd.apply(pre), LRELU_SLOPE))

    def act_relu(self, pre):
        return _RoundFwd.apply(F.relu(_RoundBwd.apply(pre)))


class MaskedPrec(Prec):
    """The reference arithmetic with the LeakyReLU DECISIONS taken from outside: the k-th activation of the forward pass uses
    masks[k] (True: slope 1, False: slope 0.2) instead of the sign of its own pre-activation.  With the masks a device run stored
    (sign of its saved activations) the network is the same piecewise-linear function the device differentiated, so its gradients
    may differ from the device's by arithmetic rounding only — which turns "the fp32x3 gradients leave the 1e-3 gate through kink
    flips only" from prose into an assertion (tests/test_gpu_baseline_shapes.py).  `flips[k]` counts the elements whose own sign
    disagrees with the mask (pre-activations at rounding level)."""
    name = "masked"

    def __init__(self, masks):
        self.masks, self.k, self.flips, self.sizes = list(masks), 0, [], []

    def act(self, pre):
        m = self.masks[self.k]
        self.k += 1
        assert m.shape == pre.shape, (self.k - 1, tuple(m.shape), tuple(pre.shape))
        self.flips.append(int(((pre.detach() > 0) != m).sum()))
        self.sizes.append(m.numel())
        one = torch.ones((), dtype=pre.dtype)
        return pre * torch.where(m, one, one * LRELU_SLOPE)      # the slope in pre's own precision, as F.leaky_relu applies it

    lrelu_raw = act


FP32 = Prec()
BF16 = _PrecBF16()


# ----------------------------------------------------------------------------
# integer index maps (bit-exact gate)
# ----------------------------------------------------------------------------
def pixel_unshuffle_index(c: int, hh: int, hw: int, s: int):
    """ssr/archs/arch_util.py:769-785 as an explicit index map:
    out[b, ch*s*s + i*s + j, y, x] = in[b, ch, y*s + i, x*s + j]."""
    h, w = hh // s, hw // s
    src = torch.empty(c * s * s, h, w, 3, dtype=torch.long)
    for ch in range(c):
        for i in range(s):
            for j in range(s):
                oc = ch * s * s + i * s + j
                ys = torch.arange(h).view(h, 1).expand(h, w)
                xs = torch.arange(w).view(1, w).expand(h, w)
                src[oc, :, :, 0] = ch
                src[oc, :, :, 1] = ys * s + i
                src[oc, :, :, 2] = xs * s + j
    return src


def pixel_unshuffle(x: torch.Tensor, s: int) -> torch.Tensor:
    b, c, hh, hw = x.shape
    assert hh % s == 0 and hw % s == 0
    idx = pixel_unshuffle_index(c, hh, hw, s)
    return x[:, idx[..., 0], idx[..., 1], idx[..., 2]]


def nearest_up2_index(n_out: int) -> torch.Tensor:
    """F.interpolate(scale_factor=2, mode='nearest') source index = floor(o/2)
    (ssr/archs/rrdbnet_arch.py:127-128)."""
    return torch.arange(n_out) // 2


def stitch_offsets(grid: int = 16, chunk: int = 128):
    """ssr/utils/infer_utils.py:41-60: chunk (i, j) is pasted at rows i*chunk, cols j*chunk."""
    return [[(i * chunk, j * chunk) for j in range(grid)] for i in range(grid)]


def quantize_u8_truncate(x: torch.Tensor) -> torch.Tensor:
    """ssr/infer_grid.py:60-64: clamp(0,1) -> *255 -> astype(uint8) (truncation, not rounding)."""
    return (x.clamp(0, 1) * 255).to(torch.uint8)


# ----------------------------------------------------------------------------
# Generator: SSR_RRDBNet (ssr/archs/rrdbnet_arch.py)
# ----------------------------------------------------------------------------
def _conv(sd, name, x, stride=1, pad=1, prec: Prec = FP32):
    return F.conv2d(x, prec.w(sd[name + ".weight"]), sd.get(name + ".bias"), stride=stride, padding=pad)


def _lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def rdb_forward(sd, pfx: str, x: torch.Tensor, prec: Prec = FP32, store: bool = True) -> torch.Tensor:
    """ResidualDenseBlock.forward, rrdbnet_arch.py:37-44.  `store=False`: the caller folds the result into a
    larger fused epilogue and rounds there (third RDB of an RRDB)."""
    feats = [x]
    for k in range(1, 5):
        feats.append(prec.act(_conv(sd, f"{pfx}.conv{k}", torch.cat(feats, 1), prec=prec)))
    x5 = _conv(sd, f"{pfx}.conv5", torch.cat(feats, 1), prec=prec)
    out = x5 * 0.2 + x
    return prec.a(out) if store else out


def rrdb_forward(sd, pfx: str, x: torch.Tensor, prec: Prec = FP32) -> torch.Tensor:
    """RRDB.forward, rrdbnet_arch.py:63-68.  (bf16 model: `(x5*0.2 + x)*0.2 + x_rrdb` is ONE epilogue, one rounding.)"""
    out = x
    for j in (1, 2, 3):
        out = rdb_forward(sd, f"{pfx}.rdb{j}", out, prec, store=(j < 3))
    return prec.a(out * 0.2 + x)


def generator_num_blocks(sd) -> int:
    n = 0
    while f"body.{n}.rdb1.conv1.weight" in sd:
        n += 1
    return n


def generator_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, scale: int = 4, prec: Prec = FP32) -> torch.Tensor:
    """SSR_RRDBNet.forward, rrdbnet_arch.py:116-137.  `sd` uses the reference's key layout."""
    a = prec.a
    x = a(x)
    if scale == 2:
        feat = pixel_unshuffle(x, 2)
    elif scale == 1:
        feat = pixel_unshuffle(x, 4)
    else:
        feat = x
    feat = a(_conv(sd, "conv_first", feat, prec=prec))
    body = feat
    for i in range(generator_num_blocks(sd)):
        body = rrdb_forward(sd, f"body.{i}", body, prec)
    feat = a(feat + _conv(sd, "conv_body", body, prec=prec))
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")   # folded into the consumer's read: never stored
    feat = prec.act(_conv(sd, "conv_up1", up(feat), prec=prec))
    feat = prec.act(_conv(sd, "conv_up2", up(feat), prec=prec))
    if scale in (8, 16):
        feat = prec.act(_conv(sd, "conv_up3", up(feat), prec=prec))
        if scale == 16:
            feat = prec.act(_conv(sd, "conv_up4", up(feat), prec=prec))
    return a(_conv(sd, "conv_last", prec.act(_conv(sd, "conv_hr", feat, prec=prec)), prec=prec))


def generator_init(num_in_ch, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                   seed: Optional[int] = None) -> "OrderedDict[str, torch.Tensor]":
    """Parameter set with the reference's key layout and init *distributions*
    (torch Conv2d default; RDB convs kaiming_normal*0.1 + zero bias, rrdbnet_arch.py:35,
    arch_util.py:600-628).  Not bit-identical to the reference's RNG consumption order — the
    golden fixtures carry the reference's actual tensors."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    if scale == 2:
        num_in_ch *= 4
    elif scale == 1:
        num_in_ch *= 16
    sd = OrderedDict()

    def default_conv(name, cin, cout, k=3):
        bound = 1.0 / math.sqrt(cin * k * k)
        sd[name + ".weight"] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def rdb_conv(name, cin, cout):
        std = math.sqrt(2.0 / (cin * 9))
        sd[name + ".weight"] = torch.randn(cout, cin, 3, 3, generator=g) * std * 0.1
        sd[name + ".bias"] = torch.zeros(cout)

    default_conv("conv_first", num_in_ch, num_feat)
    for i in range(num_block):
        for j in (1, 2, 3):
            p = f"body.{i}.rdb{j}"
            for k in range(1, 5):
                rdb_conv(f"{p}.conv{k}", num_feat + (k - 1) * num_grow_ch, num_grow_ch)
            rdb_conv(f"{p}.conv5", num_feat + 4 * num_grow_ch, num_feat)
    default_conv("conv_body", num_feat, num_feat)
    default_conv("conv_up1", num_feat, num_feat)
    default_conv("conv_up2", num_feat, num_feat)
    if scale in (8, 16):
        default_conv("conv_up3", num_feat, num_feat)
        if scale == 16:
            default_conv("conv_up4", num_feat, num_feat)
    default_conv("conv_hr", num_feat, num_feat)
    default_conv("conv_last", num_feat, num_out_ch)
    return sd


# ----------------------------------------------------------------------------
# Discriminator: SSR_UNetDiscriminatorSN (ssr/archs/discriminator_arch.py)
# ----------------------------------------------------------------------------
SN_LAYERS = tuple(f"conv{i}" for i in range(1, 9))  # discriminator_arch.py:30-39
SN_EPS = 1e-12                                      # torch.nn.utils.spectral_norm default


def spectral_norm_weight(w_orig: torch.Tensor, u: torch.Tensor, v: torch.Tensor, train: bool):
    """Old hook-style torch.nn.utils.spectral_norm as used at discriminator_arch.py:7,26.

    In training mode one power iteration is done *in place on the buffers under no_grad*:
        v <- normalize(W^T u), u <- normalize(W v), then sigma = u . (W v), W_sn = W / sigma,
    where u, v enter sigma as constants (cloned) and W carries grad.
    Returns (W_sn, u_new, v_new)."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if train:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=SN_EPS)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v


def discriminator_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, train: bool = True,
                          skip_connection: bool = True, update_buffers: bool = True, prec: Prec = FP32) -> torch.Tensor:
    """SSR_UNetDiscriminatorSN.forward, discriminator_arch.py:42-71.
    `sd` holds conv0/conv9 .weight/.bias and conv1..8 .weight_orig/.weight_u/.weight_v.
    When `train` and `update_buffers`, the u/v entries of `sd` are replaced by the updated vectors
    (that is what the in-place hook does, once per forward call)."""
    w = {}
    for name in SN_LAYERS:
        w_sn, u, v = spectral_norm_weight(sd[name + ".weight_orig"], sd[name + ".weight_u"],
                                          sd[name + ".weight_v"], train)
        if train and update_buffers:
            sd[name + ".weight_u"], sd[name + ".weight_v"] = u, v
        w[name] = prec.w(w_sn)     # W/sigma is formed in fp32 and rounded when it is packed

    a = prec.a

    def sn(name, t, stride, pad):
        return F.conv2d(t, w[name], None, stride=stride, padding=pad)

    bil = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    x0 = prec.act(F.conv2d(a(x), prec.w(sd["conv0.weight"]), sd["conv0.bias"], padding=1))
    x1 = prec.act(sn("conv1", x0, 2, 1))
    x2 = prec.act(sn("conv2", x1, 2, 1))
    x3 = prec.act(sn("conv3", x2, 2, 1))
    x3 = a(bil(x3))
    x4 = prec.act(sn("conv4", x3, 1, 1))
    if skip_connection:
        x4 = x4 + prec.g(x2)    # the sum is formed inside the interpolation's read (bf16 model: not stored; the skip
    x4 = a(bil(x4))             # branch's gradient is a buffer of its own)
    x5 = prec.act(sn("conv5", x4, 1, 1))
    if skip_connection:
        x5 = x5 + prec.g(x1)
    x5 = a(bil(x5))
    if skip_connection:         # conv6's epilogue stores lrelu(acc) + x0 with ONE rounding
        x6 = prec.w(prec.lrelu_raw(prec.g(sn("conv6", x5, 1, 1))) + prec.g(x0))
    else:
        x6 = prec.act(sn("conv6", x5, 1, 1))
    out = prec.act(sn("conv7", x6, 1, 1))
    out = prec.act(sn("conv8", out, 1, 1))
    return a(F.conv2d(out, prec.w(sd["conv9.weight"]), sd["conv9.bias"], padding=1))


def discriminator_init(num_in_ch, num_feat=64, seed: Optional[int] = None):
    """Reference-layout parameter/buffer set with torch-default init distributions; u, v are
    normalized gaussians as torch's spectral_norm creates them."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    nf = num_feat
    sd = OrderedDict()

    def conv(name, cin, cout, k, bias):
        bound = 1.0 / math.sqrt(cin * k * k)
        wgt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
        if bias:
            sd[name + ".weight"] = wgt
            sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound
        else:
            sd[name + ".weight_orig"] = wgt
            sd[name + ".weight_u"] = F.normalize(torch.randn(cout, generator=g), dim=0, eps=SN_EPS)
            sd[name + ".weight_v"] = F.normalize(torch.randn(cin * k * k, generator=g), dim=0, eps=SN_EPS)

    conv("conv0", num_in_ch, nf, 3, True)
    conv("conv1", nf, nf * 2, 4, False)
    conv("conv2", nf * 2, nf * 4, 4, False)
    conv("conv3", nf * 4, nf * 8, 4, False)
    conv("conv4", nf * 8, nf * 4, 3, False)
    conv("conv5", nf * 4, nf * 2, 3, False)
    conv("conv6", nf * 2, nf, 3, False)
    conv("conv7", nf, nf, 3, False)
    conv("conv8", nf, nf, 3, False)
    conv("conv9", nf, 1, 3, True)
    return sd


D_PARAM_KEYS = (["conv0.weight", "conv0.bias"] + [f"{n}.weight_orig" for n in SN_LAYERS]
                + ["conv9.weight", "conv9.bias"])


# ----------------------------------------------------------------------------
# Losses / optimizer pieces executed inside optimize_parameters (BasicSR 1.4.2 semantics)
# ----------------------------------------------------------------------------
def usm_gaussian_kernel1d(radius: int = 50, sigma: float = 0.0) -> torch.Tensor:
    """cv2.getGaussianKernel(ksize, sigma) as BasicSR's USMSharp.__init__ calls it (basicsr==1.4.2,
    basicsr/utils/img_process_util.py; not on disk: restated from the published source and anchored on the call site
    /root/reference/ssr/models/ssr_esrgan_model.py:31 `USMSharp().cuda()` -> radius 50 -> ksize 51, sigma 0 -> OpenCV's
    rule sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8 = 8.0; coefficients exp(-(i-c)^2 / (2 sigma^2)) normalised to sum 1)."""
    if radius % 2 == 0:
        radius += 1
    if sigma <= 0:
        sigma = 0.3 * ((radius - 1) * 0.5 - 1) + 0.8
    i = torch.arange(radius, dtype=torch.float64) - (radius - 1) / 2
    k = torch.exp(-(i * i) / (2 * sigma * sigma))
    return k / k.sum()


def filter2d(img: torch.Tensor, kernel2d: torch.Tensor) -> torch.Tensor:
    """basicsr filter2D: reflect-pad k//2 and correlate every channel of every image with the same kernel."""
    k = kernel2d.shape[-1]
    b, c, h, w = img.shape
    x = F.pad(img, (k // 2,) * 4, mode="reflect").view(b * c, 1, h + k - 1, w + k - 1)
    return F.conv2d(x, kernel2d.view(1, 1, k, k).to(img.dtype)).view(b, c, h, w)


def usm_sharp(img: torch.Tensor, weight: float = 0.5, threshold: float = 10.0, radius: int = 50, sigma: float = 0.0):
    """USMSharp.forward (applied to self.gt at ssr_esrgan_model.py:109; parity unpinned by the reference: BasicSR piece)."""
    k1 = usm_gaussian_kernel1d(radius, sigma)
    k2 = torch.outer(k1, k1).to(torch.float32)          # FloatTensor(np.dot(kernel, kernel.T))
    blur = filter2d(img, k2)
    residual = img - blur
    mask = (residual.abs() * 255 > threshold).to(img.dtype)
    soft = filter2d(mask, k2)
    sharp = torch.clip(img + weight * residual, 0, 1)
    return soft * sharp + (1 - soft) * img


# ----------------------------------------------------------------------------
# Perceptual loss: basicsr PerceptualLoss + VGGFeatureExtractor (basicsr==1.4.2, basicsr/losses/basic_loss.py and
# basicsr/archs/vgg_arch.py; not on disk: restated from the published source, parity unpinned by the reference) as configured
# by ssr/options/esrgan_s2naip_urban.yml:123-137 and called at ssr/models/ssr_esrgan_model.py:153-160.
# The network is torchvision's vgg19().features (requirements.txt:12 torchvision==0.16.0): 16 3x3 convolutions with ReLU,
# 2x2 max-pooling after conv1_2 / conv2_2 / conv3_4 / conv4_4; features are read at the NAMED layer ('convX_Y' = before the ReLU).
# ----------------------------------------------------------------------------
VGG19_LAYERS = []            # [(name, kind, torchvision features index)]
_idx = 0
for _blk, _n in enumerate((2, 2, 4, 4, 4), start=1):
    for _j in range(1, _n + 1):
        VGG19_LAYERS.append((f"conv{_blk}_{_j}", "conv", _idx)); _idx += 1
        VGG19_LAYERS.append((f"relu{_blk}_{_j}", "relu", _idx)); _idx += 1
    if _blk < 5:
        VGG19_LAYERS.append((f"pool{_blk}", "pool", _idx)); _idx += 1
VGG19_WIDTHS = {1: 64, 2: 128, 3: 256, 4: 512, 5: 512}
VGG_MEAN = (0.485, 0.456, 0.406)     # VGGFeatureExtractor.use_input_norm: the ImageNet statistics, image in [0, 1]
VGG_STD = (0.229, 0.224, 0.225)


def vgg19_init(seed: Optional[int] = None):
    """torchvision VGG._initialize_weights: kaiming_normal_(mode='fan_out', nonlinearity='relu'), zero bias; keys as in the
    torchvision checkpoint (`features.{idx}.weight`).  (The published weights vgg19-dcbb9e9d.pth cannot be fetched here:
    random weights of the same architecture for throughput and parity.)"""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    sd = OrderedDict()
    cin = 3
    for name, kind, idx in VGG19_LAYERS:
        if kind != "conv":
            continue
        cout = VGG19_WIDTHS[int(name[4])]
        sd[f"features.{idx}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cout * 9))
        sd[f"features.{idx}.bias"] = torch.zeros(cout)
        cin = cout
    return sd


def vgg19_features(sd, x: torch.Tensor, layer_names, use_input_norm: bool = True, range_norm: bool = False,
                   prec: Prec = FP32) -> Dict[str, torch.Tensor]:
    """VGGFeatureExtractor.forward: {name: activation at that layer} for the requested names."""
    if range_norm:
        x = (x + 1) / 2
    if use_input_norm:
        mean = torch.tensor(VGG_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
        std = torch.tensor(VGG_STD, dtype=x.dtype).view(1, 3, 1, 1)
        x = (x - mean) / std
    x = prec.a(x)
    want, out = set(layer_names), {}
    last = max(i for i, (n, _, _) in enumerate(VGG19_LAYERS) if n in want)
    pre = None
    for name, kind, idx in VGG19_LAYERS[:last + 1]:
        if kind == "conv":
            pre = F.conv2d(x, prec.w(sd[f"features.{idx}.weight"]), sd[f"features.{idx}.bias"], padding=1)
            if name in want:                      # a tapped conv output is stored as it is; its ReLU is applied by the consumer
                pre = prec.a(pre)
                out[name] = pre
                x = pre
            else:
                x = None
        elif kind == "relu":
            x = F.relu(x) if x is not None else prec.act_relu(pre)
            if name in want:
                out[name] = x
        else:
            x = prec.a(F.max_pool2d(x, 2, 2))
    return out


def perceptual_loss(vgg_sd, x, gt, layer_weights: Dict[str, float], perceptual_weight: float = 1.0, use_input_norm=True,
                    range_norm=False, prec: Prec = FP32):
    """PerceptualLoss.forward with criterion 'l1', style_weight 0: sum_k w_k * mean|vgg_k(x) - vgg_k(gt)| * perceptual_weight."""
    fx = vgg19_features(vgg_sd, x, layer_weights.keys(), use_input_norm, range_norm, prec)
    with torch.no_grad():
        fg = vgg19_features(vgg_sd, gt.detach(), layer_weights.keys(), use_input_norm, range_norm, prec)
    loss = 0
    for k, w in layer_weights.items():
        loss = loss + F.l1_loss(prec.g(fx[k]), fg[k]) * w
    return loss * perceptual_weight


def l1_loss(pred, target, weight=1.0):
    """basicsr L1Loss(loss_weight, reduction='mean'); call site ssr_esrgan_model.py:148."""
    return weight * F.l1_loss(pred, target, reduction="mean")


def gan_loss_vanilla(pred, target_is_real: bool, is_disc: bool, loss_weight=0.1,
                     real_label_val=1.0, fake_label_val=0.0):
    """basicsr GANLoss('vanilla') = BCEWithLogitsLoss against a constant label map;
    loss_weight applies only for the generator (is_disc=False).
    Call sites ssr_esrgan_model.py:182,218,224."""
    tv = real_label_val if target_is_real else fake_label_val
    loss = F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, tv))
    return loss if is_disc else loss * loss_weight


@dataclass
class AdamState:
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0) state, restated explicitly."""
    lr: float = 1e-4
    betas: Tuple[float, float] = (0.9, 0.99)
    eps: float = 1e-8
    step: int = 0
    exp_avg: Dict[str, torch.Tensor] = field(default_factory=dict)
    exp_avg_sq: Dict[str, torch.Tensor] = field(default_factory=dict)

    def update(self, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor]):
        self.step += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.step
        bc2 = 1 - b2 ** self.step
        for k, g in grads.items():
            if g is None:
                continue
            m = self.exp_avg.setdefault(k, torch.zeros_like(g))
            v = self.exp_avg_sq.setdefault(k, torch.zeros_like(g))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k] = params[k] - (self.lr / bc1) * (m / denom)


def model_ema(ema: Dict[str, torch.Tensor], params: Dict[str, torch.Tensor], decay: float):
    """basicsr BaseModel.model_ema: ema = ema*decay + p*(1-decay); ssr_esrgan_model.py:230-231."""
    for k in params:
        ema[k] = ema[k] * decay + params[k] * (1 - decay)


@dataclass
class StepConfig:
    """The knobs optimize_parameters reads (ssr_esrgan_model.py:119-233, esrgan_s2naip_urban.yml:96-147)."""
    l1_weight: float = 1.0
    gan_weight: float = 0.1
    lr_g: float = 1e-4
    lr_d: float = 1e-4
    betas: Tuple[float, float] = (0.9, 0.99)
    ema_decay: float = 0.999
    net_d_iters: int = 1
    net_d_init_iters: int = 0
    feed_disc_lr: bool = False
    scale: int = 4
    l1_gt_usm: bool = False        # ssr_esrgan_model.py:121-129 (the measured configuration uses the plain gt)
    gan_gt_usm: bool = False
    prec: Prec = FP32              # FP32 = the reference's arithmetic; BF16 = model of the HIP throughput mode
    percep_gt_usm: bool = False
    perceptual: Optional[Dict] = None   # train.perceptual_opt (layer_weights, perceptual_weight, use_input_norm, range_norm)


class ESRGANOracle:
    """CPU restatement of SSRESRGANModel.optimize_parameters (ssr_esrgan_model.py:119-233) with the
    L1 + vanilla-GAN losses of the measured configuration (perceptual/USM excluded).

    State is held as plain dicts in the reference's state_dict layout."""

    def __init__(self, g_sd, d_sd, cfg: StepConfig = StepConfig(), vgg_sd=None):
        self.cfg = cfg
        self.vgg = None if vgg_sd is None else OrderedDict((k, v.detach().clone().float()) for k, v in vgg_sd.items())
        assert not cfg.perceptual or self.vgg is not None, "perceptual_opt needs VGG19 weights"
        self.g = OrderedDict((k, v.detach().clone().float()) for k, v in g_sd.items())
        self.d = OrderedDict((k, v.detach().clone().float()) for k, v in d_sd.items())
        self.g_ema = OrderedDict((k, v.clone()) for k, v in self.g.items())  # model_ema(0), :49
        self.opt_g = AdamState(lr=cfg.lr_g, betas=cfg.betas)
        self.opt_d = AdamState(lr=cfg.lr_d, betas=cfg.betas)
        self.log = OrderedDict()
        self.output = None

    def _disc_input(self, img, lr_resized, old_hr=None):
        # ssr_esrgan_model.py:171-178 / :202-213: [img | lr_resized (feed_disc_lr) | old_hr (when the batch has one)]
        parts = [img]
        if self.cfg.feed_disc_lr:
            parts.append(lr_resized)
        if old_hr is not None:
            parts.append(old_hr)
        return torch.cat(parts, 1) if len(parts) > 1 else img

    def step(self, lr: torch.Tensor, gt: torch.Tensor, current_iter: int = 1, old_hr: Optional[torch.Tensor] = None):
        cfg = self.cfg
        prec = cfg.prec
        log = OrderedDict()
        with torch.no_grad():
            lr = prec.a(lr)
            gt_usm = usm_sharp(gt) if (cfg.l1_gt_usm or cfg.gan_gt_usm or (cfg.perceptual and cfg.percep_gt_usm)) else None   # :109
            l1_gt = prec.a(gt_usm if cfg.l1_gt_usm else gt)                          # :121-129
            gan_gt = prec.a(gt_usm if cfg.gan_gt_usm else gt)
            percep_gt = prec.a(gt_usm if cfg.percep_gt_usm else gt) if cfg.perceptual else None
            old_hr = prec.a(old_hr) if old_hr is not None else None
            gt = gan_gt
        lr_resized = F.interpolate(lr, scale_factor=4)  # :133 (nearest)
        # ---- optimize net_g (:136-193); D params frozen -> D contributes dgrad only
        gp = OrderedDict((k, v.detach().requires_grad_(True)) for k, v in self.g.items())
        d_frozen = OrderedDict((k, v.detach()) for k, v in self.d.items())
        output = generator_forward(gp, lr, cfg.scale, prec)
        if current_iter % cfg.net_d_iters == 0 and current_iter > cfg.net_d_init_iters:
            l_g_pix = l1_loss(prec.g(output), l1_gt, cfg.l1_weight)
            fake_g_pred = discriminator_forward(d_frozen, self._disc_input(output, lr_resized, old_hr), train=True,
                                                prec=prec)
            l_g_gan = gan_loss_vanilla(fake_g_pred, True, is_disc=False, loss_weight=cfg.gan_weight)
            l_g_total = l_g_pix + l_g_gan
            if cfg.perceptual:                                                           # :153-160
                po = cfg.perceptual
                l_g_percep = perceptual_loss(self.vgg, output, percep_gt, po["layer_weights"], float(po.get("perceptual_weight", 1.0)),
                                             po.get("use_input_norm", True), po.get("range_norm", False), prec)
                l_g_total = l_g_total + l_g_percep
                log["l_g_percep"] = l_g_percep.item()
            grads = torch.autograd.grad(l_g_total, list(gp.values()))
            log["l_g_pix"], log["l_g_gan"] = l_g_pix.item(), l_g_gan.item()
            self.g_grads = OrderedDict(zip(gp.keys(), grads))
            self.opt_g.update(self.g, self.g_grads)
            # power-iteration buffers advanced by that D forward persist (hook updates in place)
            for n in SN_LAYERS:
                self.d[n + ".weight_u"] = d_frozen[n + ".weight_u"]
                self.d[n + ".weight_v"] = d_frozen[n + ".weight_v"]
        output = output.detach()
        self.output = output
        # ---- optimize net_d (:196-228): two backward() calls accumulate into the same .grad
        dp = OrderedDict((k, (v.detach().requires_grad_(True) if k in D_PARAM_KEYS else v.detach()))
                         for k, v in self.d.items())
        real_d_pred = discriminator_forward(dp, self._disc_input(gt, lr_resized, old_hr), train=True, prec=prec)
        l_d_real = gan_loss_vanilla(real_d_pred, True, is_disc=True)
        plist = [dp[k] for k in D_PARAM_KEYS]
        g_real = torch.autograd.grad(l_d_real, plist)
        fake_d_pred = discriminator_forward(dp, self._disc_input(output, lr_resized, old_hr).detach().clone(), train=True,
                                            prec=prec)
        l_d_fake = gan_loss_vanilla(fake_d_pred, False, is_disc=True)
        g_fake = torch.autograd.grad(l_d_fake, plist)
        self.d_grads = OrderedDict((k, a + b) for k, a, b in zip(D_PARAM_KEYS, g_real, g_fake))
        log["l_d_real"], log["out_d_real"] = l_d_real.item(), real_d_pred.detach().mean().item()
        log["l_d_fake"], log["out_d_fake"] = l_d_fake.item(), fake_d_pred.detach().mean().item()
        d_params = OrderedDict((k, self.d[k]) for k in D_PARAM_KEYS)
        self.opt_d.update(d_params, self.d_grads)
        for k in D_PARAM_KEYS:
            self.d[k] = d_params[k]
        for n in SN_LAYERS:
            self.d[n + ".weight_u"] = dp[n + ".weight_u"].detach()
            self.d[n + ".weight_v"] = dp[n + ".weight_v"].detach()
        if cfg.ema_decay > 0:
            model_ema(self.g_ema, self.g, cfg.ema_decay)  # :230-231
        self.log = log
        return log


# ----------------------------------------------------------------------------
# algorithmic FLOPs (SURVEY.md §8d / BASELINE.md §3) — used by bench.py for the roofline
# ----------------------------------------------------------------------------
def generator_conv_macs(num_in_ch, H=32, W=32, nf=64, nb=23, gc=32, num_out_ch=3):
    px = H * W
    rdb = sum((nf + k * gc) * gc for k in range(4)) + (nf + 4 * gc) * nf
    macs = {"conv_first": 9 * num_in_ch * nf * px, "body": 9 * rdb * 3 * nb * px, "conv_body": 9 * nf * nf * px,
            "conv_up1": 9 * nf * nf * px * 4, "conv_up2": 9 * nf * nf * px * 16, "conv_hr": 9 * nf * nf * px * 16,
            "conv_last": 9 * nf * num_out_ch * px * 16}
    return macs


def discriminator_conv_macs(num_in_ch, H=128, W=128, nf=64):
    px = H * W
    return {"conv0": 9 * num_in_ch * nf * px, "conv1": 16 * nf * 2 * nf * px // 4,
            "conv2": 16 * 2 * nf * 4 * nf * px // 16, "conv3": 16 * 4 * nf * 8 * nf * px // 64,
            "conv4": 9 * 8 * nf * 4 * nf * px // 16, "conv5": 9 * 4 * nf * 2 * nf * px // 4,
            "conv6": 9 * 2 * nf * nf * px, "conv7": 9 * nf * nf * px, "conv8": 9 * nf * nf * px,
            "conv9": 9 * nf * px}


def step_gflop_per_image(c_in: int, c_d: int) -> float:
    """step = 3*G_fwd - 2*MAC(conv_first) + 8*D_fwd - 4*MAC(conv0)   (BASELINE.md §3), in GFLOP."""
    g = generator_conv_macs(c_in)
    d = discriminator_conv_macs(c_d)
    macs = 3 * sum(g.values()) - g["conv_first"] + 8 * sum(d.values()) - 2 * d["conv0"]
    # note: "- 2*MAC" in FLOP terms is one MAC count (2 FLOP each)
    return 2.0 * macs / 1e9 - 0.0
