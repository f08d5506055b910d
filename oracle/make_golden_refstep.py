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
the three methods read are set by hand from the same option keys.  The *_gt_usm options are False in these fixtures, so the
stand-in sharpener's output is never used.

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
    u.USMSharp = type("USMSharp", (nn.Module,), {"forward": lambda self, img: img})
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
        ema_decay=0.999, net_d_iters=1, net_d_init_iters=0):
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
        batch = {"lr": torch.randint(0, 256, (B, c_in, 8, 8), dtype=torch.uint8),
                 "hr": torch.randint(0, 256, (B, 3, 32, 32), dtype=torch.uint8)}
        if with_old_hr:
            batch["old_hr"] = torch.randint(0, 256, (B, 3, 32, 32), dtype=torch.uint8)
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
          "test_output": m.output.detach().clone(), "source": "unmodified SSRESRGANModel.feed_data / optimize_parameters / test"}
    torch.save(fx, os.path.join(OUT, name + ".pt"))
    print(name, [dict(l) for l in logs][-1])


def main():
    os.makedirs(OUT, exist_ok=True)
    g = dict(num_in_ch=6, num_out_ch=3, scale=4, num_feat=16, num_block=2, num_grow_ch=8)
    gen("stepref_plain", 11, 2, 2, g, dict(num_in_ch=3, num_feat=8, skip_connection=True), {})
    gen("stepref_feedlr_oldhr", 12, 2, 2, g, dict(num_in_ch=3 + 6 + 3, num_feat=8, skip_connection=True), {"feed_disc_lr": True},
        with_old_hr=True)
    gen("stepref_gated", 13, 2, 4, g, dict(num_in_ch=3, num_feat=8, skip_connection=True), {}, net_d_iters=2, net_d_init_iters=1)


if __name__ == "__main__":
    main()
