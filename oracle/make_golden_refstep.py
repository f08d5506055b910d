"""TEST INFRASTRUCTURE — generates tests/golden/stepref_*.pt by executing the UNMODIFIED reference methods

    SSRESRGANModel.feed_data            /root/reference/ssr/models/ssr_esrgan_model.py:104-117
    SSRESRGANModel.optimize_parameters  /root/reference/ssr/models/ssr_esrgan_model.py:119-233
    SSRESRGANModel.test                 /root/reference/ssr/models/ssr_esrgan_model.py:235-244

on the unmodified reference arch classes (SSR_RRDBNet, SSR_UNetDiscriminatorSN), on the CPU.  Round 2's step fixtures
(make_golden.py: gen_step) re-typed the control flow of optimize_parameters around those classes; here the reference's own text
runs: the freeze / unfreeze of net_d, `fake_disc_input.detach().clone()`, the two discriminator backwards, the gate
(net_d_iters / net_d_init_iters), the channel order of the discriminator input with feed_disc_lr / old_hr and the place of the EMA
update are whatever ssr_esrgan_model.py says.

The module imports BasicSR (basicsr==1.4.2, requirements.txt:1 — not installed, not in /root/reference).  Only these names are
stood in for, with BasicSR's published behaviour — the part of row a12 that stays "restated":
  basicsr.models.srgan_model.SRGANModel   base class; only model_ema() and reduce_loss_dict() are used by the methods above
  basicsr.archs.build_network, basicsr.utils.{USMSharp,get_root_logger,imwrite,tensor2img}, MODEL_REGISTRY   import-time names
  ssr.losses.build_loss / ssr.metrics.calculate_metric   import-time names (ssr/losses/__init__.py pulls clip / kornia)
  cri_pix = L1Loss(loss_weight, 'mean'), cri_gan = GANLoss('vanilla', 1.0, 0.0, loss_weight)   the two loss objects
The model object is created WITHOUT running __init__ (which calls .cuda() and builds networks through BasicSR); the attributes
the three methods read are set by hand from the same option keys.  The *_gt_usm options are False in the first three fixtures;
`stepref_usm` (round 4) runs the shipped YAMLs' setting `l1_gt_usm: True` (esrgan_s2naip_urban.yml:9-11) through the unmodified
feed_data (`self.gt_usm = self.usm_sharpener(self.gt)`, :109) and optimize_parameters (`l1_gt = self.gt_usm`, :121-129) with
the sharpener `_USMSharp` below: BasicSR's USMSharp / filter2D text (basicsr/utils/img_process_util.py, published) over a
restatement of the one OpenCV call it makes, cv2.getGaussianKernel(51, 0) (cv2 is not installed: OpenCV's documented rule, in
double precision as OpenCV computes a CV_64F kernel).  Which target gets the sharpened image, and where it enters the losses, is
the reference's text; the sharpener is BasicSR's published text; only getGaussianKernel is a formula.

    python -m oracle.make_golden_refstep          (build container only: needs /root/reference)
"""
import copy
import importlib
import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.ref_shim import REFERENCE_ROOT, load_reference_archs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _SRGANModelStandIn:
    """The two BaseModel / SRModel methods optimize_parameters calls (BasicSR 1.4.2, published behaviour)."""

    def model_ema(self, decay=0.999):            # SRModel.model_ema
        net_g_params = dict(self.net_g.named_parameters())
        net_g_ema_params = dict(self.net_g_ema.named_parameters())
        for k in net_g_ema_params.keys():
            net_g_ema_params[k].data.mul_(decay).add_(net_g_params[k].data, alpha=1 - decay)

    def reduce_loss_dict(self, loss_dict):       # BaseModel.reduce_loss_dict without a process group
        with torch.no_grad():
            log_dict = OrderedDict()
            for name, value in loss_dict.items():
                log_dict[name] = value.mean().item()
            return log_dict


class _L1Loss(nn.Module):                         # basicsr.losses.L1Loss(loss_weight, reduction='mean')
    def __init__(self, loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, target):
        return self.loss_weight * F.l1_loss(pred, target, reduction="mean")


class _GANLoss(nn.Module):                        # basicsr.losses.GANLoss('vanilla', real_label_val=1.0, fake_label_val=0.0, loss_weight)
    def __init__(self, loss_weight=1.0, real_label_val=1.0, fake_label_val=0.0):
        super().__init__()
        self.loss_weight, self.real_label_val, self.fake_label_val = loss_weight, real_label_val, fake_label_val
        self.loss = nn.BCEWithLogitsLoss()

    def forward(self, input, target_is_real, is_disc=False):
        target = input.new_ones(input.size()) * (self.real_label_val if target_is_real else self.fake_label_val)
        loss = self.loss(input, target)
        return loss if is_disc else loss * self.loss_weight      # loss_weight only for the generator


def _cv2_get_gaussian_kernel(ksize: int, sigma: float):
    """cv2.getGaussianKernel(ksize, sigma, ktype=CV_64F) for ksize > 7 or sigma > 0 (OpenCV imgproc/smooth: sigma <= 0 ->
    0.3*((ksize-1)*0.5 - 1) + 0.8; coefficients exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)) in double, scaled to sum 1).  Returns
    the (ksize, 1) float64 numpy column OpenCV returns."""
    import numpy as np
    assert ksize % 2 == 1 and (ksize > 7 or sigma > 0), "the small fixed tables of OpenCV are not restated"
    sigma_x = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    scale2x = -0.5 / (sigma_x * sigma_x)
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(scale2x * x * x)
    return (k * (1.0 / k.sum())).reshape(ksize, 1)


def _filter2D(img, kernel):                       # basicsr.utils.img_process_util.filter2D (1.4.2), same-kernel branch
    k = kernel.size(-1)
    b, c, h, w = img.size()
    if k % 2 == 1:
        img = F.pad(img, (k // 2, k // 2, k // 2, k // 2), mode="reflect")
    else:
        raise ValueError("Wrong kernel size")
    ph, pw = img.size()[-2:]
    assert kernel.size(0) == 1
    img = img.view(b * c, 1, ph, pw)
    kernel = kernel.view(1, 1, k, k)
    return F.conv2d(img, kernel, padding=0).view(b, c, h, w)


class _USMSharp(nn.Module):                       # basicsr.utils.img_process_util.USMSharp (1.4.2)
    def __init__(self, radius=50, sigma=0):
        super().__init__()
        import numpy as np
        if radius % 2 == 0:
            radius += 1
        self.radius = radius
        kernel = _cv2_get_gaussian_kernel(radius, sigma)
        kernel = torch.FloatTensor(np.dot(kernel, kernel.transpose())).unsqueeze_(0)
        self.register_buffer("kernel", kernel)

    def forward(self, img, weight=0.5, threshold=10):
        blur = _filter2D(img, self.kernel)
        residual = img - blur
        mask = torch.abs(residual) * 255 > threshold
        mask = mask.float()
        soft_mask = _filter2D(mask, self.kernel)
        sharp = img + weight * residual
        sharp = torch.clip(sharp, 0, 1)
        return soft_mask * sharp + (1 - soft_mask) * img


_CACHE = {}


def load_reference_model_class():
    """(the unmodified SSRESRGANModel class object, the reference arch classes, the stand-in sharpener class)"""
    if _CACHE:
        return _CACHE["v"]
    G, D, _ = load_reference_archs()
    import basicsr  # the stand-in package of ref_shim
    archs = types.ModuleType("basicsr.archs")
    archs.build_network = lambda opt: (_ for _ in ()).throw(NotImplementedError("not used: __init__ is bypassed"))
    models = types.ModuleType("basicsr.models")
    srgan = types.ModuleType("basicsr.models.srgan_model")
    srgan.SRGANModel = _SRGANModelStandIn
    models.srgan_model = srgan
    basicsr.archs, basicsr.models = archs, models
    u = sys.modules["basicsr.utils"]
    u.USMSharp = _USMSharp
    u.imwrite = lambda *a, **k: None
    u.tensor2img = lambda *a, **k: None
    sys.modules.update({"basicsr.archs": archs, "basicsr.models": models, "basicsr.models.srgan_model": srgan})
    losses = types.ModuleType("ssr.losses")
    losses.build_loss = lambda opt: (_ for _ in ()).throw(NotImplementedError("not used: init_training_settings is bypassed"))
    metrics = types.ModuleType("ssr.metrics")
    metrics.calculate_metric = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    pkg = types.ModuleType("ssr.models")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "ssr", "models")]      # skip ssr/models/__init__.py (it imports every model file)
    sys.modules.update({"ssr.losses": losses, "ssr.metrics": metrics, "ssr.models": pkg})
    _CACHE["v"] = (importlib.import_module("ssr.models.ssr_esrgan_model").SSRESRGANModel, G, D, u.USMSharp)
    return _CACHE["v"]


def sd_of(net):
    return OrderedDict((k, v.detach().clone()) for k, v in net.state_dict().items())


def gen(name, seed, B, n_iters, g_kw, d_kw, opt_extra, with_old_hr=False, l1_w=1.0, gan_w=0.1, lr=1e-4, betas=(0.9, 0.99),
        ema_decay=0.999, net_d_iters=1, net_d_init_iters=0, lr_hw=8, smooth_hr=False):
    Model, G, D, Sharp = load_reference_model_class()
    torch.manual_seed(seed)
    m = object.__new__(Model)                      # no __init__: the three methods below are what is exercised
    m.device = torch.device("cpu")
    m.opt = dict({"l1_gt_usm": False, "percep_gt_usm": False, "gan_gt_usm": False}, **opt_extra)
    m.net_g, m.net_d = G(**g_kw).train(), D(**d_kw).train()
    m.net_g_ema = copy.deepcopy(m.net_g).eval()    # init_training_settings :43-49 (model_ema(0) = a copy)
    m.usm_sharpener = Sharp()
    m.cri_pix, m.cri_gan = _L1Loss(l1_w), _GANLoss(gan_w)
    m.cri_ldl = m.cri_perceptual = m.ssim_loss = m.clip_sim = None
    m.net_d_iters, m.net_d_init_iters, m.ema_decay = net_d_iters, net_d_init_iters, ema_decay
    m.optimizer_g = torch.optim.Adam(m.net_g.parameters(), lr=lr, weight_decay=0, betas=betas)
    m.optimizer_d = torch.optim.Adam(m.net_d.parameters(), lr=lr, weight_decay=0, betas=betas)
    g0, d0 = sd_of(m.net_g), sd_of(m.net_d)
    data, logs, g_grads1, d_grads1 = [], [], None, None
    c_in = g_kw["num_in_ch"]
    for it in range(1, n_iters + 1):
        batch = {"lr": torch.randint(0, 256, (B, c_in, lr_hw, lr_hw), dtype=torch.uint8),
                 "hr": torch.randint(0, 256, (B, 3, 4 * lr_hw, 4 * lr_hw), dtype=torch.uint8)}
        if smooth_hr:      # an image with structure: white noise puts |residual| * 255 > 10 almost everywhere and the soft mask near 1
            base = F.interpolate(torch.rand(B, 3, lr_hw // 2, lr_hw // 2), scale_factor=8, mode="bicubic", align_corners=False)
            batch["hr"] = (base.clamp(0, 1) * 255 + torch.randint(-6, 7, base.shape)).clamp(0, 255).round().to(torch.uint8)
        if with_old_hr:
            batch["old_hr"] = torch.randint(0, 256, (B, 3, 4 * lr_hw, 4 * lr_hw), dtype=torch.uint8)
        data.append(batch)
        m.feed_data(batch)                         # UNMODIFIED :104-117
        m.optimize_parameters(it)                  # UNMODIFIED :119-233
        if g_grads1 is None and any(p.grad is not None for p in m.net_g.parameters()):
            g_grads1 = (it, OrderedDict((n, p.grad.detach().clone()) for n, p in m.net_g.named_parameters()))
        if it == 1:
            d_grads1 = OrderedDict((n, p.grad.detach().clone()) for n, p in m.net_d.named_parameters())
        logs.append(OrderedDict(m.log_dict))
    m.test()                                       # UNMODIFIED :235-244 (net_g_ema under no_grad)
    fx = {"g_kwargs": g_kw, "d_kwargs": d_kw, "opt": m.opt, "with_old_hr": with_old_hr, "l1_weight": l1_w, "gan_weight": gan_w,
          "lr": lr, "betas": betas, "ema_decay": ema_decay, "net_d_iters": net_d_iters, "net_d_init_iters": net_d_init_iters,
          "g0": g0, "d0": d0, "data": data, "logs": logs, "g_grads_first": g_grads1, "d_grads_iter1": d_grads1,
          "g_final": sd_of(m.net_g), "d_final": sd_of(m.net_d), "g_ema_final": sd_of(m.net_g_ema),
          "test_output": m.output.detach().clone(), "gt_usm_last": m.gt_usm.detach().clone() if hasattr(m, "gt_usm") else None,
          "source": "unmodified SSRESRGANModel.feed_data / optimize_parameters / test"}
    torch.save(fx, os.path.join(OUT, name + ".pt"))
    print(name, [dict(l) for l in logs][-1])


def main():
    os.makedirs(OUT, exist_ok=True)
    g = dict(num_in_ch=6, num_out_ch=3, scale=4, num_feat=16, num_block=2, num_grow_ch=8)
    gen("stepref_plain", 11, 2, 2, g, dict(num_in_ch=3, num_feat=8, skip_connection=True), {})
    gen("stepref_feedlr_oldhr", 12, 2, 2, g, dict(num_in_ch=3 + 6 + 3, num_feat=8, skip_connection=True), {"feed_disc_lr": True},
        with_old_hr=True)
    gen("stepref_gated", 13, 2, 4, g, dict(num_in_ch=3, num_feat=8, skip_connection=True), {}, net_d_iters=2, net_d_init_iters=1)
    # the shipped setting (esrgan_s2naip_urban.yml:9-11): L1 against the USM-sharpened ground truth, GAN against the plain one;
    # 64 x 64 targets (reflect padding needs 25 < H)
    gen("stepref_usm", 14, 2, 2, g, dict(num_in_ch=3, num_feat=8, skip_connection=True), {"l1_gt_usm": True, "gan_gt_usm": False},
        lr_hw=16, smooth_hr=True)


if __name__ == "__main__":
    main()
