"""VGG19 perceptual loss on MI355X: forward of the feature extractor on the generator output and on the target, feature-L1
losses, and the backward to the generator output — BasicSR's PerceptualLoss / VGGFeatureExtractor as configured by
/root/reference/ssr/options/esrgan_s2naip_urban.yml:123-137 and called at /root/reference/ssr/models/ssr_esrgan_model.py:153-160
(`l_g_percep, l_g_style = self.cri_perceptual(self.output, percep_gt)`; style_weight 0 -> no style term).

VGG19 (torchvision `features`, frozen): 16 3x3 convolutions + ReLU, 2x2 max-pool after conv1_2 / conv2_2 / conv3_4 / conv4_4; the
loss reads the conv outputs BEFORE the ReLU.  The convolutions and their dgrads run on the conv kernels through the C ABI
(SSR_ACT_RELU epilogue, ReLU' masks); the tapped layers are exactly the ones in front of a pooling, so "ReLU + max-pool" is one
pass over the stored pre-ReLU feature (csrc/vgg.hip).  Only dgrads are needed (the extractor's parameters are frozen).
The gradient w.r.t. the generator output is ADDED to the L1-gradient buffer that the discriminator's input-gradient kernel
already folds in (train_step._phase_g), so no extra pass over the 128x128 output exists.

Weights: torchvision's checkpoint layout (`features.{idx}.weight/bias`, e.g. vgg19-dcbb9e9d.pth; BasicSR's `vgg_net.` prefix is
accepted too).  There is no network access here, so benchmarks and tests use random weights of the same architecture."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch

from . import engine, hip
from .hip import view

VGG_PRETRAIN_PATH = "experiments/pretrained_models/vgg19-dcbb9e9d.pth"    # basicsr/archs/vgg_arch.py
VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)
# (block, convs in the block, width)
VGG19_BLOCKS = ((1, 2, 64), (2, 2, 128), (3, 4, 256), (4, 4, 512), (5, 4, 512))


def vgg19_layers():
    """[(name 'convB_J', torchvision features index, cin, cout, pooled_before)]"""
    out, idx, cin = [], 0, 3
    for b, n, width in VGG19_BLOCKS:
        for j in range(1, n + 1):
            out.append((f"conv{b}_{j}", idx, cin, width, j == 1 and b > 1))
            idx += 2                    # conv, relu
            cin = width
        idx += 1                        # pool
    return out


def vgg19_specs(last: str) -> List[engine.ConvSpec]:
    specs = []
    for name, idx, cin, cout, _ in vgg19_layers():
        specs.append(engine.ConvSpec(f"features.{idx}", cout, cin, 3, 1, True, False))
        if name == last:
            break
    return specs


def vgg19_random_state(specs, seed: int = 0):
    """torchvision VGG._initialize_weights: kaiming_normal_(fan_out, relu), zero bias."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for s in specs:
        sd[s.name + ".weight"] = torch.randn(s.cout, s.cin, 3, 3, generator=g) * math.sqrt(2.0 / (s.cout * 9))
        sd[s.name + ".bias"] = torch.zeros(s.cout)
    return sd


def load_vgg19_state(path: str) -> Dict[str, torch.Tensor]:
    sd = torch.load(path, map_location="cpu")
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    return {k[len("vgg_net."):] if k.startswith("vgg_net.") else k: v for k, v in sd.items() if "features." in k}


class PerceptualPlan:
    """Static launch lists for one (B, H, W): `fwd_target` (features of the target, no activations kept), `fwd` (features of
    the generator output, activations kept), `bwd` (loss, feature gradients, dgrads down to the image)."""

    def __init__(self, opt: Dict, B: int, H: int, W: int, dtype: int, x_buf: torch.Tensor, tgt_buf: torch.Tensor,
                 grad_buf: torch.Tensor, loss_ptr: int, num_ch: int = 3, state: Optional[Dict[str, torch.Tensor]] = None,
                 loss_flags: int = 0):
        if opt.get("type", "PerceptualLoss") != "PerceptualLoss":
            raise NotImplementedError(f"train.perceptual_opt.type={opt.get('type')!r}")
        if opt.get("vgg_type", "vgg19") != "vgg19":
            raise NotImplementedError(f"train.perceptual_opt.vgg_type={opt.get('vgg_type')!r}: only vgg19")
        if float(opt.get("style_weight", 0)) != 0.0:
            raise NotImplementedError("train.perceptual_opt.style_weight != 0 (Gram-matrix style loss)")
        if opt.get("criterion", "l1") != "l1":
            raise NotImplementedError(f"train.perceptual_opt.criterion={opt.get('criterion')!r}: only l1")
        assert num_ch == 3 and H % 16 == 0 and W % 16 == 0, "VGG19 features need RGB images with H, W divisible by 16"
        self.layer_weights = {str(k): float(v) for k, v in opt["layer_weights"].items()}
        self.pw = float(opt.get("perceptual_weight", 1.0))
        names = [n for n, *_ in vgg19_layers()]
        for k in self.layer_weights:
            if k not in names:
                raise NotImplementedError(f"perceptual layer {k!r}: features are supported at conv outputs (before the ReLU)")
        layers = vgg19_layers()
        last = max(self.layer_weights, key=names.index)
        layers = layers[:names.index(last) + 1]
        for (n, *_), nxt in zip(layers, layers[1:]):
            # a tapped layer is stored before its ReLU; the fused ReLU+pool pass needs a pooling behind it
            if n in self.layer_weights and not nxt[4]:
                raise NotImplementedError(f"perceptual layer {n!r}: taps are supported in front of a pooling layer (and at the last layer)")
        mode = dtype                          # (hip.F32F: the VGG forward in exact fp32 too - its ReLU decisions shape the image gradient)
        dtype = hip.storage_code(mode)
        self.dt, self.B, self.H, self.W = dtype, B, H, W
        tdt = hip.torch_dtype(dtype)
        dev = x_buf.device
        self.store = engine.ParamStore(vgg19_specs(last), mode, device=dev)
        sd = state
        self.random_weights = False
        if sd is None:
            # BasicSR's VGGFeatureExtractor: experiments/pretrained_models/vgg19-dcbb9e9d.pth if present, else torchvision's
            # download.  There is no download here: the file (torchvision layout) must exist, or the caller passes `state`.
            cands = [opt.get("pretrained_path"), os.environ.get("SSR_VGG19_WEIGHTS"), VGG_PRETRAIN_PATH]
            path = next((c for c in cands if c and os.path.exists(c)), None)
            if path is not None:
                sd = load_vgg19_state(path)
            elif os.environ.get("SSR_VGG19_RANDOM") == "1":   # explicit opt-in: throughput runs without the published weights
                sd = vgg19_random_state(list(self.store.specs.values()), seed=int(opt.get("seed", 0)))
                self.random_weights = True
            else:
                raise FileNotFoundError(
                    f"perceptual_opt needs the VGG19 weights: put torchvision's vgg19-dcbb9e9d.pth at {VGG_PRETRAIN_PATH} (BasicSR's "
                    "location), or set SSR_VGG19_WEIGHTS=<file>; SSR_VGG19_RANDOM=1 runs with random weights (throughput only)")
        self.store.load_state_dict(sd)
        z = lambda *s: torch.zeros(*s, dtype=tdt, device=dev)
        use_norm, range_norm = bool(opt.get("use_input_norm", True)), bool(opt.get("range_norm", False))
        # y = ((x + 1)/2 if range_norm else x - mean)/std ...   as one per-channel affine map
        a = 0.5 if range_norm else 1.0
        b = 0.5 if range_norm else 0.0
        mean, std = (VGG_MEAN, VGG_STD) if use_norm else ((0.0,) * 3, (1.0,) * 3)
        self._scale = (C.c_float * 8)(*[a / s for s in std], *([0.0] * 5))
        self._shift = (C.c_float * 8)(*[(b - m) / s for m, s in zip(mean, std)], *([0.0] * 5))
        self._zero = (C.c_float * 8)(*([0.0] * 8))
        self.xn = z(B, H, W, 8)          # normalised image (3 of 8 channels)
        self.g_xn = z(B, H, W, 8)
        cb = engine._ConvBuilder(self.store, B)
        self._cb = cb
        lib = hip.lib()
        # activations: per conv either the post-ReLU output (plain layers) or the pre-ReLU feature (tapped layers) + its pooled twin
        acts, feats_t, pooled, g_acts, g_pooled = {}, {}, {}, {}, {}
        h, w = H, W
        dims = {}
        for name, idx, cin, cout, pooled_before in layers:
            if pooled_before:
                h, w = h // 2, w // 2
            dims[name] = (h, w)
            acts[name] = z(B, h, w, cout)
            g_acts[name] = z(B, h, w, cout)
            if name in self.layer_weights:
                feats_t[name] = z(B, h, w, cout)
                if name != last:
                    pooled[name] = z(B, h // 2, w // 2, cout)
                    g_pooled[name] = z(B, h // 2, w // 2, cout)
        self.acts, self.feats_t, self.g_acts = acts, feats_t, g_acts
        npix = B * H * W

        def forward(img: torch.Tensor, outs: Dict[str, torch.Tensor]) -> engine.Launcher:
            L = engine.Launcher()
            L.add(lib.ssr_channel_affine, view(img), view(self.xn), dtype, npix, 3, self._scale, self._shift, 0, what="vgg normalise")
            src = self.xn
            for name, idx, cin, cout, pooled_before in layers:
                hh, ww = dims[name]
                tapped = name in self.layer_weights
                dst = outs[name] if tapped else acts[name]
                cb.conv(L, f"features.{idx}", view(src), hh, ww, view(dst), act=hip.ACT_NONE if tapped else hip.ACT_RELU)
                src = dst
                if tapped and name != last:
                    L.add(lib.ssr_relu_maxpool2_fwd, view(dst), view(pooled[name]), dtype, B, hh, ww, cout, what=f"relu+pool {name}")
                    src = pooled[name]
            return L

        self.fwd_target = forward(tgt_buf, feats_t)
        self.fwd = forward(x_buf, acts)
        # ---- backward: losses + feature gradients, then dgrads from the last layer down to the image
        Bk = engine.Launcher()
        for name, wgt in self.layer_weights.items():
            hh, ww = dims[name]
            cout = acts[name].shape[-1]
            # loss += w_k * pw * mean|Fx - Ft| ; g_F = w_k * pw * sign(Fx - Ft) / numel
            Bk.add(lib.ssr_l1_loss, view(acts[name]), view(feats_t[name]), view(g_acts[name]), dtype | loss_flags, B * hh * ww, cout, wgt * self.pw,
                   loss_ptr, what=f"feature L1 {name}")
        for li in reversed(range(len(layers))):
            name, idx, cin, cout, pooled_before = layers[li]
            hh, ww = dims[name]
            if li == 0:      # conv1_1: gradient w.r.t. the normalised image (3 channels), no mask
                cb.dgrad(Bk, f"features.{idx}", view(g_acts[name]), hh, ww, view(self.g_xn), cout=8, cin_dy=cout)
                break
            pname = layers[li - 1][0]
            if pooled_before:
                # input is pooled[pname]: plain dgrad into its gradient, then route through ReLU + pooling into the tapped feature
                cb.dgrad(Bk, f"features.{idx}", view(g_acts[name]), hh, ww, view(g_pooled[pname]))
                ph, pw_ = dims[pname]
                Bk.add(lib.ssr_relu_maxpool2_bwd, view(acts[pname]), view(g_pooled[pname]), view(g_acts[pname]), dtype, B, ph, pw_,
                       acts[pname].shape[-1], 1, what=f"relu+pool bwd {pname}")
            else:
                # input is the ReLU output of a plain layer: ReLU' from the sign of the stored activation
                assert pname not in self.layer_weights
                cb.dgrad(Bk, f"features.{idx}", view(g_acts[name]), hh, ww, view(g_acts[pname]), m=view(acts[pname]), m_c0=0,
                         m_c1=acts[pname].shape[-1], m_relu=1)
        # d loss / d image = d loss / d xn * scale, added to the generator's output-gradient residual
        self._gscale = (C.c_float * 8)(*[float(self._scale[c]) for c in range(3)], *([0.0] * 5))
        Bk.add(lib.ssr_channel_affine, view(self.g_xn), view(grad_buf), dtype, npix, 3, self._gscale, self._zero, 1, what="vgg normalise bwd")
        self.bwd = Bk
        self.layers, self.dims, self.pooled, self.g_pooled = layers, dims, pooled, g_pooled

    def pack(self):
        self.store.pack()
