"""SSRESRGANModel on MI355X — the model plugin the training loop drives
(/root/reference/ssr/train.py:65,106-133), mirroring /root/reference/ssr/models/ssr_esrgan_model.py:

    feed_data(:104-117)  optimize_parameters(:119-233)  test(:235-244)  get_current_visuals(:246-252)
    + the BasicSR BaseModel methods train.py calls: update_learning_rate, get_current_learning_rate,
      get_current_log, save, resume_training.

Same registry name and the same `opt` dictionary (YAML) keys.  Instead of autograd over ~41k ATen ops per
step, optimize_parameters() replays the fused HIP-graph step of train_step.ESRGANTrainStep.
Scope: L1 + vanilla-GAN losses.  Options that need components outside the hot path
(perceptual/VGG, CLIP, SSIM losses, old_hr) raise NotImplementedError; USM-sharpened ground truth (l1_gt_usm /
gan_gt_usm, the shipped YAML's setting) runs on the GPU (csrc/misc.hip usm_sharp_kernel)
instead of being silently ignored."""
from __future__ import annotations

import os
from collections import OrderedDict

import torch

from .. import hip
from ..dp import init_distributed
from ..registry import MODEL_REGISTRY
from ..train_step import ESRGANTrainStep, StepConfig


def _arch_kwargs(net_opt: dict, expect: str) -> dict:
    kw = {k: v for k, v in net_opt.items() if k != "type"}
    if net_opt.get("type", expect) != expect:
        raise NotImplementedError(f"network type {net_opt.get('type')!r}: only {expect} is on the MI355X hot path")
    return kw


@MODEL_REGISTRY.register()
class SSRESRGANModel:
    def __init__(self, opt: dict):
        self.opt = opt
        self.is_train = opt.get("is_train", True)
        if not torch.cuda.is_available():
            raise hip.HipLibraryError("SSRESRGANModel needs an MI355X (no CPU fallback)")
        hip.lib()
        self.device = torch.device("cuda")
        self.dp = init_distributed() if opt.get("dist", False) else None
        self.g_kwargs = _arch_kwargs(opt["network_g"], "SSR_RRDBNet")
        self.d_kwargs = _arch_kwargs(opt["network_d"], "SSR_UNetDiscriminatorSN")
        self.compute_dtype = self.g_kwargs.pop("compute_dtype", opt.get("compute_dtype", "fp32"))
        self.d_kwargs.pop("compute_dtype", None)
        self.feed_disc_lr = bool(opt.get("feed_disc_lr", False))
        train_opt = opt.get("train", {})
        for k in ("perceptual_opt", "ldl_opt", "ssim_opt", "clip_opt"):
            if train_opt.get(k):
                raise NotImplementedError(f"train.{k}: outside the MI355X hot path (SURVEY.md §8f)")
        pix, gan = train_opt.get("pixel_opt") or {}, train_opt.get("gan_opt") or {}
        if gan and gan.get("gan_type", "vanilla") != "vanilla":
            raise NotImplementedError("only gan_type: vanilla")
        og, od = train_opt.get("optim_g", {}), train_opt.get("optim_d", {})
        self.cfg = StepConfig(
            l1_weight=float(pix.get("loss_weight", 1.0)), gan_weight=float(gan.get("loss_weight", 0.1)),
            lr_g=float(og.get("lr", 1e-4)), lr_d=float(od.get("lr", 1e-4)),
            betas=tuple(og.get("betas", (0.9, 0.99))), ema_decay=float(train_opt.get("ema_decay", 0)),
            net_d_iters=int(train_opt.get("net_d_iters", 1)), net_d_init_iters=int(train_opt.get("net_d_init_iters", 0)),
            feed_disc_lr=self.feed_disc_lr,
            # `if self.opt['l1_gt_usm'] is False: l1_gt = self.gt` (ssr_esrgan_model.py:121-129): sharpened unless exactly False
            l1_gt_usm=opt.get("l1_gt_usm", False) is not False, gan_gt_usm=opt.get("gan_gt_usm", False) is not False,
            real_label=float(gan.get("real_label_val", 1.0)),
            fake_label=float(gan.get("fake_label_val", 0.0)))
        sch = train_opt.get("scheduler", {})
        self.milestones = list(sch.get("milestones", []))
        self.gamma = float(sch.get("gamma", 1.0))
        self.ts = None
        self._pending_state = None
        self.log_dict = OrderedDict()
        self.lr = self.gt = self.output = None
        self.current_lrs = [self.cfg.lr_g, self.cfg.lr_d]

    # ---- plan creation is lazy: shapes come from the first batch ----
    def _ensure(self, B, h, w):
        if self.ts is None or (self.ts.B, self.ts.h, self.ts.w) != (B, h, w):
            old = self.ts
            self.ts = ESRGANTrainStep(self.g_kwargs, self.d_kwargs, B, h, w, self.compute_dtype, self.cfg, dp=self.dp,
                                      g_store=old.g_store if old else None, d_store=old.d_store if old else None)
            if old is None:
                self._init_params()
            else:  # carry optimizer state over to the new shape
                for a, b in ((self.ts.opt_g, old.opt_g), (self.ts.opt_d, old.opt_d)):
                    a.exp_avg.copy_(b.exp_avg); a.exp_avg_sq.copy_(b.exp_avg_sq); a.step.copy_(b.step)
                if old.opt_g.ema is not None:
                    self.ts.opt_g.ema.copy_(old.opt_g.ema)

    def _init_params(self):
        from ..archs.rrdbnet_arch import SSR_RRDBNet
        from ..archs.discriminator_arch import SSR_UNetDiscriminatorSN
        path = self.opt.get("path", {})
        seed = self.opt.get("manual_seed")
        if seed is not None:
            torch.manual_seed(int(seed))
        g_sd = SSR_RRDBNet(**self.g_kwargs).state_dict()
        d_sd = SSR_UNetDiscriminatorSN(**self.d_kwargs).state_dict()
        if path.get("pretrain_network_g"):   # load_network(net_g, path, strict_load_g, param_key_g)
            ck = torch.load(path["pretrain_network_g"], map_location="cpu")
            g_sd = ck[path.get("param_key_g", "params")]
        if path.get("pretrain_network_d"):
            ck = torch.load(path["pretrain_network_d"], map_location="cpu")
            d_sd = ck[path.get("param_key_d", "params")]
        self.ts.load_state(g_sd, d_sd)
        if self.dp is not None:
            self.ts.sync_params_from_rank0()

    # ---- the methods train.py calls ----
    def feed_data(self, data: dict):
        """ssr_esrgan_model.py:104-117: uint8 tensors -> float/255 on the device."""
        if "old_hr" in data:
            raise NotImplementedError("old_hr discriminator input: outside the measured hot path")
        lr = data["lr"].to(self.device, non_blocking=True).float()
        B, _, h, w = lr.shape
        self._ensure(B, h, w)
        self.lr = lr / 255
        if "hr" in data:
            gt = data["hr"].to(self.device, non_blocking=True).float()
            self.gt = gt / 255
            self.ts.feed_data(lr, gt, scale=1.0 / 255)
        else:
            self.ts.g_plan.load_input(lr.contiguous(), 1.0 / 255)

    def optimize_parameters(self, current_iter: int):
        # the reference indexes these keys directly (KeyError when absent): ssr_esrgan_model.py:124-129
        _ = [self.opt["l1_gt_usm"], self.opt["percep_gt_usm"], self.opt["gan_gt_usm"]]   # applied in feed_data (StepConfig)
        self.ts.step(current_iter)
        self.output = None  # materialised lazily by get_current_visuals()

    def get_current_log(self):
        self.log_dict = self.ts.log()
        return self.log_dict

    def update_learning_rate(self, current_iter: int, warmup_iter: int = -1):
        lr_g, lr_d = self.cfg.lr_g, self.cfg.lr_d
        k = sum(1 for m in self.milestones if current_iter >= m)      # MultiStepLR (esrgan_s2naip_urban.yml:109-112)
        f = self.gamma ** k
        if warmup_iter > 0 and current_iter < warmup_iter:            # BasicSR linear warm-up
            f *= current_iter / warmup_iter
        self.current_lrs = [lr_g * f, lr_d * f]
        if self.ts is not None:
            self.ts.opt_g.set_lr(self.current_lrs[0])
            self.ts.opt_d.set_lr(self.current_lrs[1])

    def get_current_learning_rate(self):
        return [self.current_lrs[0]]

    def test(self):
        """:235-244 — forward with the EMA weights under no_grad."""
        from .. import engine
        ts = self.ts
        st = engine.ParamStore(engine.generator_specs(**self.g_kwargs), ts.dt)
        st.data.copy_(ts.opt_g.ema if ts.opt_g.ema is not None else ts.g_store.data)
        plan = engine.GeneratorPlan(st, ts.B, ts.h, ts.w, training=False, **self.g_kwargs)
        st.pack()
        plan.load_input(self.lr.contiguous())
        plan.fwd.run()
        self.output = plan.read_output()

    def get_current_visuals(self):
        out = OrderedDict()
        out["lr"] = self.lr.detach().cpu()
        out["result"] = (self.output if self.output is not None else self.ts.output()).detach().cpu()
        if self.gt is not None:
            out["gt"] = self.gt.detach().cpu()
        return out

    def save(self, epoch: int, current_iter: int):
        """BasicSR layout: net_g_{iter}.pth = {'params', 'params_ema'}, net_d_{iter}.pth = {'params'},
        training_states/{iter}.state."""
        if self.dp is not None and self.dp.rank != 0:
            return
        it = "latest" if current_iter == -1 else str(current_iter)
        path = self.opt.get("path", {})
        mdir, sdir = path.get("models", "experiments/models"), path.get("training_states", "experiments/training_states")
        os.makedirs(mdir, exist_ok=True)
        os.makedirs(sdir, exist_ok=True)
        cpu = lambda sd: OrderedDict((k, v.cpu()) for k, v in sd.items())
        g = {"params": cpu(self.ts.g_store.state_dict())}
        if self.ts.opt_g.ema is not None:
            g["params_ema"] = cpu(self.ts.ema_state_dict())
        torch.save(g, os.path.join(mdir, f"net_g_{it}.pth"))
        torch.save({"params": cpu(self.ts.d_store.state_dict())}, os.path.join(mdir, f"net_d_{it}.pth"))
        state = {"epoch": epoch, "iter": current_iter,
                 "optimizers": [{"exp_avg": o.exp_avg.cpu(), "exp_avg_sq": o.exp_avg_sq.cpu(), "step": int(o.step.item())}
                                for o in (self.ts.opt_g, self.ts.opt_d)]}
        torch.save(state, os.path.join(sdir, f"{it}.state"))

    def resume_training(self, resume_state: dict):
        for o, s in zip((self.ts.opt_g, self.ts.opt_d), resume_state["optimizers"]):
            o.exp_avg.copy_(s["exp_avg"]); o.exp_avg_sq.copy_(s["exp_avg_sq"]); o.step.fill_(s["step"])

    def validation(self, dataloader, current_iter, tb_logger, save_img=False):
        raise NotImplementedError("validation metrics are outside the hot path (SURVEY.md §8f rank 4)")
