"""SSRESRGANModel on MI355X — the model plugin the training loop drives
(/root/reference/ssr/train.py:62-65,106-133), mirroring /root/reference/ssr/models/ssr_esrgan_model.py:

    feed_data(:104-117)  optimize_parameters(:119-233)  test(:235-244)  get_current_visuals(:246-252)
    nondist_validation(:269-352)
    + the BasicSR BaseModel methods train.py calls: update_learning_rate, get_current_learning_rate,
      get_current_log, save, resume_training, validation.

Same registry name and the same `opt` dictionary (YAML) keys.  Instead of autograd over ~41k ATen ops per
step, optimize_parameters() replays the fused HIP-graph step of train_step.ESRGANTrainStep.
Scope: L1 + vanilla-GAN (+ VGG19 perceptual) losses, USM-sharpened ground truth, `feed_disc_lr`
and `old_hr` discriminator inputs.  Anything this path does not implement raises NotImplementedError naming the
option (clip / ssim / ldl losses, other optimizers or schedulers, weight decay) — nothing is silently ignored."""
from __future__ import annotations

import math
import os
from collections import Counter, OrderedDict

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


def step_config_from_opt(opt: dict) -> StepConfig:
    """The knobs init_training_settings reads (ssr_esrgan_model.py:33-102; esrgan_s2naip_urban.yml:96-147), validated:
    an option this path cannot honour raises instead of being dropped."""
    train_opt = opt.get("train", {})
    for k in ("ldl_opt", "ssim_opt", "clip_opt"):
        if train_opt.get(k):
            raise NotImplementedError(f"train.{k}: outside the MI355X hot path (SURVEY.md §8f)")
    pix, gan = train_opt.get("pixel_opt"), train_opt.get("gan_opt") or {}
    if pix and pix.get("type", "L1Loss") != "L1Loss":
        raise NotImplementedError(f"train.pixel_opt.type={pix.get('type')!r}: only L1Loss")
    if pix and pix.get("reduction", "mean") != "mean":
        raise NotImplementedError("train.pixel_opt.reduction: only 'mean'")
    if gan.get("type", "GANLoss") != "GANLoss" or gan.get("gan_type", "vanilla") != "vanilla":
        raise NotImplementedError("train.gan_opt: only GANLoss with gan_type: vanilla")
    og, od = dict(train_opt.get("optim_g", {})), dict(train_opt.get("optim_d", {}))
    for name, o in (("optim_g", og), ("optim_d", od)):
        if o.get("type", "Adam") != "Adam":
            raise NotImplementedError(f"train.{name}.type={o.get('type')!r}: only Adam")
        if float(o.get("weight_decay", 0)) != 0.0:
            raise NotImplementedError(f"train.{name}.weight_decay != 0")
        if o.get("amsgrad", False):
            raise NotImplementedError(f"train.{name}.amsgrad")
    sch = train_opt.get("scheduler", {})
    if sch and sch.get("type", "MultiStepLR") not in ("MultiStepLR", "MultiStepRestartLR"):
        raise NotImplementedError(f"train.scheduler.type={sch.get('type')!r}: only MultiStepLR / MultiStepRestartLR")
    if sch and (list(sch.get("restarts", [0])) != [0] or list(sch.get("restart_weights", [1])) != [1]):
        raise NotImplementedError("train.scheduler restarts")
    perc = train_opt.get("perceptual_opt")
    return StepConfig(
        l1_weight=float(pix.get("loss_weight", 1.0)) if pix else 0.0,    # no pixel_opt: cri_pix = None (:47-50)
        gan_weight=float(gan.get("loss_weight", 0.1)),
        lr_g=float(og.get("lr", 1e-4)), lr_d=float(od.get("lr", 1e-4)),
        betas=tuple(og.get("betas", (0.9, 0.999))), betas_d=tuple(od.get("betas", (0.9, 0.999))),
        eps=float(og.get("eps", 1e-8)), ema_decay=float(train_opt.get("ema_decay", 0)),
        net_d_iters=int(train_opt.get("net_d_iters", 1)), net_d_init_iters=int(train_opt.get("net_d_init_iters", 0)),
        feed_disc_lr=bool(opt.get("feed_disc_lr", False)),
        # `if self.opt['l1_gt_usm'] is False: l1_gt = self.gt` (ssr_esrgan_model.py:121-129): sharpened unless exactly False
        l1_gt_usm=opt.get("l1_gt_usm", False) is not False, gan_gt_usm=opt.get("gan_gt_usm", False) is not False,
        percep_gt_usm=opt.get("percep_gt_usm", False) is not False,
        real_label=float(gan.get("real_label_val", 1.0)), fake_label=float(gan.get("fake_label_val", 0.0)),
        perceptual=dict(perc) if perc else None,
        deterministic=bool(opt.get("deterministic", False)))   # ours: fixed-order reductions only (bit-identical runs; train_step.StepConfig)


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
        # default arithmetic: fp32h - fp16-split forward (22 bits per operand: outputs and LeakyReLU decisions of an fp32 evaluation),
        # split-bf16 backward: the fastest mode inside EVERY gate of the reference (outputs and parameter gradients at 1e-3, hip.F32H);
        # `compute_dtype: fp32f` has the forward in exact fp32 (no fp16 range: activations beyond 65504), `fp32` is exact throughout,
        # `fp32x3` (outputs inside the gate) / `bf16` are the modes outside the gradient gate
        self.compute_dtype = self.g_kwargs.pop("compute_dtype", opt.get("compute_dtype", "fp32h"))
        self.feed_disc_lr = bool(opt.get("feed_disc_lr", False))
        if self.is_train:      # test.py builds the model with is_train=False: generator only (SRGANModel.__init__ / init_training_settings)
            self.d_kwargs = _arch_kwargs(opt["network_d"], "SSR_UNetDiscriminatorSN")
            self.d_kwargs.pop("compute_dtype", None)
            self.cfg = step_config_from_opt(opt)
        else:
            self.d_kwargs, self.cfg = None, StepConfig()
        sch = opt.get("train", {}).get("scheduler", {})
        self.milestones = list(sch.get("milestones", []))
        self.gamma = float(sch.get("gamma", 0.1)) if sch else 1.0
        self.ts = None
        self._pending_state = None         # resume_training() before the first batch built the step (train.py:64-65)
        self._infer = None                 # (store, plan) of test()
        self._validating = False
        self.log_dict = OrderedDict()
        self.lr = self.gt = self.output = None
        self.current_lrs = [self.cfg.lr_g, self.cfg.lr_d]
        self.metric_results = {}
        self.best_metric_results = {}

    # ---- plan creation is lazy: shapes come from the first batch ----
    def _ensure(self, B, h, w):
        if self.ts is not None and (self.ts.B, self.ts.h, self.ts.w) == (B, h, w):
            return
        old, self.ts, self._infer = self.ts, None, None
        carry = None
        if old is not None:     # a ragged last batch / another tile size: carry every piece of state over, then free the old step
            carry = dict(g=old.g_store.state_dict(), d=old.d_store.state_dict(),
                         opt=[(o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step.clone()) for o in (old.opt_g, old.opt_d)],
                         ema=None if old.opt_g.ema is None else old.opt_g.ema.clone(), it=old.iter)
            del old
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        self.ts = ESRGANTrainStep(self.g_kwargs, self.d_kwargs, B, h, w, self.compute_dtype, self.cfg, dp=self.dp)
        if carry is None:
            self._init_params()
        else:
            self.ts.load_state(carry["g"], carry["d"], reset_ema=False)
            for o, (m, v, s) in zip((self.ts.opt_g, self.ts.opt_d), carry["opt"]):
                o.exp_avg.copy_(m); o.exp_avg_sq.copy_(v); o.step.copy_(s)
            if carry["ema"] is not None:
                self.ts.opt_g.ema.copy_(carry["ema"])
            self.ts.iter = carry["it"]
        if self._pending_state is not None:
            self._apply_resume_state(self._pending_state)
            self._pending_state = None
        # the schedule lives on the host: a freshly built step starts at the base LR
        self.ts.opt_g.set_lr(self.current_lrs[0])
        self.ts.opt_d.set_lr(self.current_lrs[1])

    def _init_params(self):
        from ..archs.rrdbnet_arch import SSR_RRDBNet
        from ..archs.discriminator_arch import SSR_UNetDiscriminatorSN
        path = self.opt.get("path", {})
        seed = self.opt.get("manual_seed")
        # initialise under a forked RNG: the caller's global stream (seeded manual_seed + rank by options.py:81 and consumed
        # by the dataloaders) is left untouched
        with torch.random.fork_rng(devices=[]):
            if seed is not None:
                torch.manual_seed(int(seed))
            g_sd = SSR_RRDBNet(**self.g_kwargs).state_dict()
            d_sd = SSR_UNetDiscriminatorSN(**self.d_kwargs).state_dict()
        ema_sd = None
        if path.get("pretrain_network_g"):   # load_network(net_g, path, strict_load_g, param_key_g) + net_g_ema <- 'params_ema'
            ck = torch.load(path["pretrain_network_g"], map_location="cpu")
            g_sd = ck[path.get("param_key_g", "params")]
            ema_sd = ck.get("params_ema")     # init_training_settings: net_g_ema loads 'params_ema' (:39-49)
        if path.get("pretrain_network_d"):
            ck = torch.load(path["pretrain_network_d"], map_location="cpu")
            d_sd = ck[path.get("param_key_d", "params")]
        self.ts.load_state(g_sd, d_sd)
        if ema_sd is not None and self.ts.opt_g.ema is not None:
            for key in self.ts.g_store.offsets:
                self.ts.g_store.tensor(key, self.ts.opt_g.ema).copy_(ema_sd[key].to(self.device, torch.float32))
        if self.dp is not None:
            self.ts.sync_params_from_rank0(reset_ema=ema_sd is None)

    # ---- the methods train.py calls ----
    def feed_data(self, data: dict):
        """ssr_esrgan_model.py:104-117: uint8 tensors -> float/255 on the device."""
        lr = data["lr"].to(self.device, non_blocking=True).float()
        B, _, h, w = lr.shape
        if not self.is_train or (self._validating and self.ts is not None):
            # validation batches (usually of another size) only feed test(): the train step, its plans and graphs stay as they are
            self.lr = lr / 255
            self.gt = data["hr"].to(self.device, non_blocking=True).float() / 255 if "hr" in data else None
            return
        has_old = "old_hr" in data and "hr" in data
        if self.ts is None:
            self.cfg.old_hr = has_old      # the D input width is static: decided by the first training batch
        self._ensure(B, h, w)
        self.lr = lr / 255
        if "hr" in data:
            gt = data["hr"].to(self.device, non_blocking=True).float()
            self.gt = gt / 255
            old = data["old_hr"].to(self.device, non_blocking=True).float() if self.cfg.old_hr else None
            self.ts.feed_data(lr, gt, scale=1.0 / 255, old_hr=old)
        else:
            self.gt = None
            self.ts.g_plan.load_input(lr.contiguous(), 1.0 / 255)

    def optimize_parameters(self, current_iter: int):
        # the reference indexes these keys directly (KeyError when absent): ssr_esrgan_model.py:124-129
        _ = [self.opt["l1_gt_usm"], self.opt["percep_gt_usm"], self.opt["gan_gt_usm"]]   # applied in feed_data (StepConfig)
        self.ts.step(current_iter)
        self.output = None  # materialised lazily by get_current_visuals()

    def get_current_log(self):
        self.log_dict = self.ts.log()
        # the fp16-split forward of mode fp32h has fp16's range (include/ssr_hip.h, SSR_F32H): an activation beyond 65504 turns into NaN outputs.
        # The losses are read from the device here anyway: say what to do instead of logging NaN silently
        if self.compute_dtype == "fp32h" and any(isinstance(v, float) and not math.isfinite(v) for v in self.log_dict.values()):
            raise FloatingPointError("non-finite loss in compute_dtype fp32h: its forward convolutions run on fp16-split operands (|activation| < 65504, "
                                     "|weight| < 64); set `compute_dtype: fp32f` (exact fp32 forward, same gates) for this model")
        return self.log_dict

    def update_learning_rate(self, current_iter: int, warmup_iter: int = -1):
        """BasicSR BaseModel.update_learning_rate: the schedulers are stepped once per call from the second call on
        (last_epoch = current_iter - 1), MultiStepRestartLR multiplies by gamma when last_epoch reaches a milestone
        (esrgan_s2naip_urban.yml:109-112); during warm-up the LR is base * current_iter / warmup_iter."""
        k = sum(1 for m in self.milestones if current_iter - 1 >= m)
        f = self.gamma ** k
        if warmup_iter > 0 and current_iter < warmup_iter:
            f = current_iter / warmup_iter
        self.current_lrs = [self.cfg.lr_g * f, self.cfg.lr_d * f]
        if self.ts is not None:
            self.ts.opt_g.set_lr(self.current_lrs[0])
            self.ts.opt_d.set_lr(self.current_lrs[1])

    def get_current_learning_rate(self):
        return [self.current_lrs[0]]

    def test(self):
        """:235-244 — forward with the EMA weights (net_g_ema) under no_grad."""
        from .. import engine
        ts = self.ts
        B, _, h, w = self.lr.shape
        key = (B, h, w)
        dt = ts.mode if ts is not None else hip.dtype_code(self.compute_dtype)      # the arithmetic MODE (fp32f: this forward is exact fp32)
        if self._infer is None or self._infer[0] != key:
            st = self._infer[1] if self._infer is not None else engine.ParamStore(engine.generator_specs(**self.g_kwargs), dt)
            if self._infer is None and ts is None:
                st.load_state_dict(self._test_weights())
            self._infer = (key, st, engine.GeneratorPlan(st, B, h, w, training=False, **self.g_kwargs))
        _, st, plan = self._infer
        if ts is not None:
            st.data.copy_(ts.opt_g.ema if ts.opt_g.ema is not None else ts.g_store.data)
        st.pack()
        plan.load_input(self.lr.contiguous(), 1.0)      # self.lr = lr/255 already (one rounding, as :107 of the reference)
        plan.fwd.run()
        self.output = plan.read_output()

    def _test_weights(self):
        """is_train=False (test.py): load_network(net_g, pretrain_network_g, strict_load_g, param_key_g); without a checkpoint the
        default initialisation under manual_seed, as constructing the reference model would give."""
        path = self.opt.get("path", {}) or {}
        if path.get("pretrain_network_g"):
            ck = torch.load(path["pretrain_network_g"], map_location="cpu")
            return ck[path.get("param_key_g", "params")]
        from ..archs.rrdbnet_arch import SSR_RRDBNet
        with torch.random.fork_rng(devices=[]):
            if self.opt.get("manual_seed") is not None:
                torch.manual_seed(int(self.opt["manual_seed"]))
            return SSR_RRDBNet(**self.g_kwargs).state_dict()

    def get_current_visuals(self):
        out = OrderedDict()
        out["lr"] = self.lr.detach().cpu()
        out["result"] = (self.output if self.output is not None else self.ts.output()).detach().cpu()
        if self.gt is not None:
            out["gt"] = self.gt.detach().cpu()
        return out

    # ---- checkpoints (BasicSR layout, loadable by either side) ----
    def _optimizer_state_dict(self, o, lr, betas):
        """torch.optim.Adam.state_dict() over the parameters in named_parameters() order (BasicSR setup_optimizers)."""
        st = o.store
        keys = list(st.offsets)
        step = float(o.step.item())
        state = {i: {"step": torch.tensor(step), "exp_avg": st.tensor(k, o.exp_avg).detach().cpu().clone(),
                     "exp_avg_sq": st.tensor(k, o.exp_avg_sq).detach().cpu().clone()} for i, k in enumerate(keys)}
        group = {"lr": lr, "betas": tuple(betas), "eps": self.cfg.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "initial_lr": lr,
                 "params": list(range(len(keys)))}
        return {"state": state, "param_groups": [group]}

    def _scheduler_state_dict(self, base_lr, lr, current_iter):
        """basicsr MultiStepRestartLR.state_dict() (= the scheduler's __dict__ minus the optimizer)."""
        last = max(int(current_iter) - 1, 0)
        return {"milestones": Counter(self.milestones), "gamma": self.gamma, "restarts": [0], "restart_weights": [1],
                "base_lrs": [base_lr], "last_epoch": last, "_step_count": last + 1, "verbose": False,
                "_get_lr_called_within_step": False, "_last_lr": [lr]}

    def save(self, epoch: int, current_iter: int):
        """BasicSR layout: net_g_{iter}.pth = {'params', 'params_ema'}, net_d_{iter}.pth = {'params'},
        training_states/{iter}.state = {'epoch', 'iter', 'optimizers': [Adam state_dicts], 'schedulers': [...]}."""
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
        cfg = self.cfg
        state = {"epoch": epoch, "iter": current_iter,
                 "optimizers": [self._optimizer_state_dict(self.ts.opt_g, self.current_lrs[0], cfg.betas),
                                self._optimizer_state_dict(self.ts.opt_d, self.current_lrs[1], cfg.betas_d or cfg.betas)],
                 "schedulers": [self._scheduler_state_dict(cfg.lr_g, self.current_lrs[0], current_iter),
                                self._scheduler_state_dict(cfg.lr_d, self.current_lrs[1], current_iter)]}
        torch.save(state, os.path.join(sdir, f"{it}.state"))

    def resume_training(self, resume_state: dict):
        """BaseModel.resume_training: optimizers and schedulers from a `.state` file.  train.py:64-65 calls it right after
        build_model(), before any batch: the step (whose shapes come from the first batch) does not exist yet, so the state
        is kept and applied when it is built."""
        assert len(resume_state["optimizers"]) == 2, "Wrong lengths of optimizers"
        if self.ts is None:
            self._pending_state = resume_state
        else:
            self._apply_resume_state(resume_state)
        it = int(resume_state.get("iter", 0))
        if it > 0:   # the schedule is a function of the iteration: restore the LR the next update_learning_rate would continue from
            self.update_learning_rate(it)

    def _apply_resume_state(self, resume_state: dict):
        for o, s in zip((self.ts.opt_g, self.ts.opt_d), resume_state["optimizers"]):
            st = o.store
            keys = list(st.offsets)
            per = s["state"]
            assert len(per) in (0, len(keys)), f"optimizer state has {len(per)} entries, network has {len(keys)} parameters"
            step = 0
            for i, k in enumerate(keys):
                if i not in per:
                    continue
                e = per[i]
                st.tensor(k, o.exp_avg).copy_(e["exp_avg"].to(self.device, torch.float32))
                st.tensor(k, o.exp_avg_sq).copy_(e["exp_avg_sq"].to(self.device, torch.float32))
                step = int(float(e["step"]))
            o.step.fill_(step)
        self.ts.iter = int(resume_state.get("iter", 0))

    # ---- validation (ssr_esrgan_model.py:269-352; metrics: psnr / ssim as BasicSR, cpsnr as ssr/metrics/cpsnr.py) ----
    def validation(self, dataloader, current_iter, tb_logger, save_img=False):
        """BaseModel.validation -> nondist_validation on every rank's own loader (the reference's dist_validation only runs
        on rank 0)."""
        if self.dp is not None and self.dp.rank != 0:
            return
        self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img):
        from .. import metrics as M
        ds_opt = getattr(getattr(dataloader, "dataset", None), "opt", None) or {}
        dataset_name = ds_opt.get("name", "val")
        sect = self.opt.get("test" if dataset_name == "test" else "val") or {}
        metrics2run = sect.get("metrics") or {}
        for name, mo in metrics2run.items():
            if mo.get("type") not in M.METRICS:
                raise NotImplementedError(f"metric {name}: type {mo.get('type')!r} is outside the MI355X path "
                                          f"(have {sorted(M.METRICS)})")
        self.metric_results = {m: 0.0 for m in metrics2run}
        rec = self.best_metric_results.setdefault(dataset_name, {})
        for m, mo in metrics2run.items():
            better = mo.get("better", "higher")
            rec.setdefault(m, dict(better=better, val=float("-inf") if better == "higher" else float("inf"), iter=-1))
        n = 0
        self._validating = True
        try:       # an exception below (bad image, unsupported metric option, OOM) must not leave the model in validation mode
            for idx, val_data in enumerate(dataloader):
                self.feed_data(val_data)
                self.test()
                sr = M.tensor2img_u8(self.output)                 # [B,H,W,3] uint8 RGB on the device (tensor2img: clamp, *255, round)
                gt = M.tensor2img_u8(self.gt) if self.gt is not None else None
                if save_img:
                    vis = self.opt.get("path", {}).get("visualization", "experiments/visualization")
                    base = os.path.join(vis, str(idx)) if self.opt.get("is_train", True) else os.path.join(vis, dataset_name)
                    os.makedirs(base, exist_ok=True)
                    tag = f"{idx}_{current_iter}" if self.opt.get("is_train", True) else f"{idx}_{self.opt.get('name', 'run')}"
                    M.imwrite_rgb(sr[0].cpu().numpy(), os.path.join(base, tag + ".png"))
                    if gt is not None:
                        M.imwrite_rgb(gt[0].cpu().numpy(), os.path.join(base, tag + "_gt.png"))
                if gt is not None:
                    for name, mo in metrics2run.items():
                        kw = {k: v for k, v in mo.items() if k not in ("type", "better")}
                        self.metric_results[name] += float(M.METRICS[mo["type"]](sr[:1], gt[:1], **kw))
                n += 1
                self.gt = self.output = None
        finally:
            self._validating = False
        for name in self.metric_results:
            self.metric_results[name] /= max(n, 1)
            r = rec[name]
            v = self.metric_results[name]
            if (r["better"] == "higher" and v >= r["val"]) or (r["better"] != "higher" and v <= r["val"]):
                r["val"], r["iter"] = v, current_iter
        if tb_logger is not None:
            for name, v in self.metric_results.items():
                tb_logger.add_scalar(f"metrics/{dataset_name}/{name}", v, current_iter)
        return self.metric_results
