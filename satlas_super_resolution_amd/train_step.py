"""The fused G+D ESRGAN train step on one MI355X (one process per GPU; see dp.py for the exchange).

Follows /root/reference/ssr/models/ssr_esrgan_model.py:119-233 (optimize_parameters) operation by
operation — generator phase with D frozen (:136-193), discriminator real/fake phases (:196-228), EMA
(:230-231) — with the L1 + vanilla-GAN losses of the measured configuration.
Everything runs through libssr_hip.so; the whole step is static and is replayed from hipGraphs.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import engine, hip
from .dp import DPContext
from .hip import AdamArgs, view


@dataclass
class StepConfig:
    l1_weight: float = 1.0         # pixel_opt.loss_weight           (esrgan_s2naip_urban.yml:118-121)
    gan_weight: float = 0.1        # gan_opt.loss_weight             (:139-144)
    lr_g: float = 1e-4             # optim_g                         (:98-102)
    lr_d: float = 1e-4             # optim_d                         (:103-107)
    betas: Tuple[float, float] = (0.9, 0.99)
    betas_d: Optional[Tuple[float, float]] = None   # optim_d.betas (None: same as optim_g's)
    eps: float = 1e-8
    ema_decay: float = 0.999       # train.ema_decay                 (:97)
    net_d_iters: int = 1           # (:146)
    net_d_init_iters: int = 0      # (:147)
    feed_disc_lr: bool = False     # top-level feed_disc_lr          (:14)
    old_hr: bool = False           # batches carry 'old_hr' (an older high-res image of the same place) for D (ssr_esrgan_model.py:112-114)
    l1_gt_usm: bool = False        # L1 target = USM-sharpened gt    (ssr_esrgan_model.py:121-125; yml :9)
    gan_gt_usm: bool = False       # D real input = USM-sharpened gt (:127-129; yml :11)
    percep_gt_usm: bool = False    # perceptual target = USM-sharpened gt (:125-126; yml :10)
    perceptual: Optional[Dict] = None   # train.perceptual_opt (VGG19 feature L1, :153-160; yml :123-137)
    real_label: float = 1.0
    fake_label: float = 0.0
    deterministic: bool = False    # fixed-order reductions only: two runs give bit-identical parameters (engine.deterministic; SSR_DETERMINISTIC=1)


LOSS_KEYS = ("l_g_pix", "l_g_gan", "l_d_real", "out_d_real", "l_d_fake", "out_d_fake")


class AdamState:
    """torch.optim.Adam state over a ParamStore's flat arena (+ optional EMA arena)."""

    def __init__(self, store: engine.ParamStore, lr: float, betas, eps: float, ema_decay: float = 0.0):
        dev = store.device
        self.store = store
        self.exp_avg = torch.zeros_like(store.data)
        self.exp_avg_sq = torch.zeros_like(store.data)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr = torch.full((1,), lr, dtype=torch.float32, device=dev)
        self.ema = store.data.clone() if ema_decay > 0 else None      # model_ema(0): copy of net_g
        self.args = AdamArgs(store.data.data_ptr(), store.grad.data_ptr(), self.exp_avg.data_ptr(),
                             self.exp_avg_sq.data_ptr(), self.ema.data_ptr() if self.ema is not None else None,
                             store.numel, self.lr.data_ptr(), self.step.data_ptr(), betas[0], betas[1], eps,
                             ema_decay, 1.0)

    def set_lr(self, lr: float):
        self.lr.fill_(lr)

    def update(self, grad_scale: float = 1.0):
        self.args.grad_scale = grad_scale
        hip.check(hip.lib().ssr_adam_step(C.byref(self.args), hip.stream_ptr()), "ssr_adam_step")


class ESRGANTrainStep:
    """State + launch plans of one rank's G+D train step for fixed (B, h, w)."""

    def __init__(self, g_kwargs: Dict, d_kwargs: Dict, B: int, h: int, w: int, dtype="fp32",
                 cfg: StepConfig = StepConfig(), dp: Optional[DPContext] = None, use_graph: bool = True,
                 g_store: Optional[engine.ParamStore] = None, d_store: Optional[engine.ParamStore] = None,
                 vgg_state: Optional[Dict[str, torch.Tensor]] = None):
        assert g_kwargs.get("scale", 4) == 4, "the train step is defined for scale 4 (all shipped configs)"
        self.cfg, self.B, self.h, self.w = cfg, B, h, w
        self.mode = hip.dtype_code(dtype)          # arithmetic mode of the launch plans (hip.F32F: exact fp32 forward, split-bf16 backward)
        self.dt = hip.storage_code(self.mode)      # what the C ABI sees for tensors, losses, backward launches
        self.dp = dp if dp is not None else DPContext(None, 0, 1)
        self.use_graph = use_graph
        self.g_kwargs, self.d_kwargs = dict(g_kwargs), dict(d_kwargs)
        cin, cout = g_kwargs["num_in_ch"], g_kwargs.get("num_out_ch", 3)
        cd = d_kwargs["num_in_ch"]
        assert cd == cout + (cin if cfg.feed_disc_lr else 0) + (cout if cfg.old_hr else 0), \
            f"network_d.num_in_ch={cd} must be {cout} (+{cin} with feed_disc_lr, +{cout} with old_hr): ssr_esrgan_model.py:171-178"
        self.cin, self.cout, self.cd = cin, cout, cd
        self.g_store = g_store or engine.ParamStore(engine.generator_specs(**g_kwargs), self.mode)
        self.d_store = d_store or engine.ParamStore(
            engine.discriminator_specs(cd, d_kwargs.get("num_feat", 64), in_hw=(4 * h, 4 * w), dtype=hip.forward_code(self.mode)), self.mode)
        tdt, dev = hip.torch_dtype(self.dt), self.g_store.device
        H, W = 4 * h, 4 * w
        self.H, self.W = H, W
        cdp = engine.rup(cd, 8)
        z = lambda *s: torch.zeros(*s, dtype=tdt, device=dev)
        self.fake_in = z(B, H, W, cdp)      # [G output | lr_resized | old_hr]   (ssr_esrgan_model.py:171-178)
        self.real_in = z(B, H, W, cdp)      # [gt       | lr_resized | old_hr]   (:202-213)
        self.grad_l1 = z(B, H, W, cdp)
        # targets (ssr_esrgan_model.py:121-129): each of the L1 / perceptual / GAN-real targets is the ground truth or its
        # USM-sharpened version; at most two distinct images exist, the D-real buffer holds the GAN one
        by_flag = {bool(cfg.gan_gt_usm): self.real_in}

        def target_for(flag: bool) -> torch.Tensor:
            if flag not in by_flag:
                by_flag[flag] = z(B, H, W, cdp)
            return by_flag[flag]
        self.l1_tgt = target_for(bool(cfg.l1_gt_usm))
        self.percep_tgt = target_for(bool(cfg.percep_gt_usm)) if cfg.perceptual else None
        self._tgt_by_flag = by_flag
        self._gt_usm = None   # fp32 NCHW scratch for ssr_usm_sharp
        # deterministic mode (cfg.deterministic / SSR_DETERMINISTIC=1): every loss scalar is SSR_LOSS_SLOTS per-block slots that
        # log() adds in index order; weight gradients of split layers go through per-split partial buffers (engine.WgradBatch);
        # the generator runs as ONE chain (two half-batch chains would add into the same gradients concurrently)
        self.det = bool(cfg.deterministic) or engine.deterministic()
        # plans capture the mode while THEY are built; the process-global switch is restored even when a constructor raises
        with engine.deterministic_mode(self.det):
            self.loss_dt = self.dt | (hip.DETERMINISTIC if self.det else 0)
            self.loss_stride = hip.LOSS_SLOTS if self.det else 1
            self.losses = torch.zeros(8 * self.loss_stride, dtype=torch.float32, device=dev)
            self.d_plan = engine.DiscriminatorPlan(self.d_store, B, H, W, num_in_ch=cd,
                                                   num_feat=d_kwargs.get("num_feat", 64),
                                                   skip_connection=d_kwargs.get("skip_connection", True))
            import os
            # the generator as two half-batch launch chains (engine.SplitGeneratorPlan) where its launches are latency chains of
            # single-wave-per-CU kernels: the fused dense blocks (bf16, nf = 64, gc = 32) at batches of two full rounds or more
            # r02g, two boxes, B = 32 8xS2 bf16: 14.40 -> 13.97 ms and 13.78 -> 13.53 ms per step with two chains; four chains
            # (half-chip launches): 14.9 ms — slower; B = 16 (one round per launch already): no difference
            # r03: the 8 x 16-tile dense-block kernel (csrc/rdb_tile.hip) wants the WHOLE batch in one launch (one workgroup per CU at
            # B = 32, 32 x 32 tiles) and brings its own second wave per SIMD; measured 12.49 ms (one chain, new kernel) vs 12.68 (two
            # chains, new kernel) vs 13.32 (two chains, 8 x 8 kernel).  "auto" = split only where the 8 x 8 kernel will run.
            env_split = os.environ.get("SSR_G_SPLIT", "auto")
            probe = hip.RdbDesc()       # ask the library which dense-block kernel this launch shape gets instead of restating its rule
            probe.dtype, probe.N, probe.H, probe.W = hip.BF16, B, h, w
            wide = hip.lib().ssr_rdb_tile_of(C.byref(probe)) == 16
            # r06d (fp32x3, the register-tiled body kernel csrc/conv_x3r.hip: one workgroup per CU whatever the batch): two half-batch chains
            # 27.00 -> 26.57 ms per step, same call - the chains' launch gaps and prologues fill each other's idle CUs
            n_split = (2 if self.dt == hip.F32X3 else (1 if wide else 2)) if env_split == "auto" else int(env_split)
            split = n_split > 1 and not self.det and self.dt in (hip.BF16, hip.F32X3) and B % n_split == 0 and B // n_split >= 16 \
                and g_kwargs.get("num_feat", 64) == 64 and g_kwargs.get("num_grow_ch", 32) == 32
            if split:
                self.g_plan = engine.SplitGeneratorPlan(self.g_store, B, h, w, training=True, out_buf=self.fake_in, d_out_buf=self.d_plan.g_in,
                                                        parts=n_split, **g_kwargs)
            else:
                # data parallel: G's backward in segments, so that a segment's slice of the gradient arena is on the wire while the next
                # segment computes (step(): one all-reduce per segment).  SSR_DP_SEGMENTS=1: one exchange behind the whole backward.
                n_seg = max(1, int(os.environ.get("SSR_DP_SEGMENTS", "3"))) if (dp is not None and dp.active) else 1
                self.g_plan = engine.GeneratorPlan(self.g_store, B, h, w, training=True, out_buf=self.fake_in, d_out_buf=self.d_plan.g_in,
                                                   bwd_segments=n_seg, **g_kwargs)
            self.p_plan = None
            if cfg.perceptual:      # VGG19 feature L1 (ssr_esrgan_model.py:153-160); its image gradient joins the L1 gradient buffer
                from .perceptual import PerceptualPlan
                self.p_plan = PerceptualPlan(cfg.perceptual, B, H, W, self.mode, self.fake_in, self.percep_tgt, self.grad_l1,
                                             self.losses.data_ptr() + 4 * 6 * self.loss_stride, num_ch=cout, state=vgg_state,
                                             loss_flags=self.loss_dt & hip.DETERMINISTIC)
                self.p_plan.pack()
            self.opt_g = AdamState(self.g_store, cfg.lr_g, cfg.betas, cfg.eps, cfg.ema_decay)
            self.opt_d = AdamState(self.d_store, cfg.lr_d, cfg.betas_d or cfg.betas, cfg.eps, 0.0)
        self._graphs: Dict[str, torch.cuda.CUDAGraph] = {}
        self._warm = set()
        self._side = None
        import os
        # r02c, same box, B=32 8xS2 bf16: 13.94 -> 13.27 ms per step with the fork (two pairs, +-0.02)
        self.overlap_d = os.environ.get("SSR_OVERLAP_D", "1") == "1"
        self.dp_fork = os.environ.get("SSR_DP_FORK", "1") == "1"
        self.dp_one_graph = os.environ.get("SSR_DP_ONE_GRAPH", "0") == "1"
        self.iter = 0

    # ------------------------------------------------------------------ state
    def load_state(self, g_sd, d_sd, reset_ema: bool = True):
        self.g_store.load_state_dict(g_sd)
        self.d_store.load_state_dict(d_sd)
        if reset_ema and self.opt_g.ema is not None:
            self.opt_g.ema.copy_(self.g_store.data)

    def ema_state_dict(self):
        out = OrderedDict()
        for key in self.g_store.offsets:
            out[key] = self.g_store.tensor(key, self.opt_g.ema).clone()
        return out

    def sync_params_from_rank0(self, reset_ema: bool = True):
        """DDP constructor semantics."""
        for t in (self.g_store.data, self.d_store.data, *self.d_store.u.values(), *self.d_store.v.values()):
            self.dp.broadcast_(t)
        if self.opt_g.ema is not None:
            if reset_ema:
                self.opt_g.ema.copy_(self.g_store.data)
            else:
                self.dp.broadcast_(self.opt_g.ema)

    # ------------------------------------------------------------------ data
    def feed_data(self, lr: torch.Tensor, gt: torch.Tensor, scale: float = 1.0, old_hr: Optional[torch.Tensor] = None):
        """lr: [B,Cin,h,w], gt (and old_hr): [B,3,4h,4w] float32 NCHW on the GPU (`scale` = 1/255 for uint8-valued
        inputs: ssr_esrgan_model.py:106-108,112-114)."""
        assert (old_hr is not None) == self.cfg.old_hr, "old_hr must be fed iff StepConfig.old_hr (the D input width is static)"
        L = hip.lib()
        st = hip.stream_ptr()
        lr = lr.contiguous()
        gt = gt.contiguous()
        self.g_plan.load_input(lr, scale)
        gt_usm = None
        if True in self._tgt_by_flag:   # self.gt_usm = self.usm_sharpener(self.gt)  (:109)
            if self._gt_usm is None:
                self._gt_usm = torch.empty(self.B, self.cout, self.H, self.W, dtype=torch.float32, device=gt.device)
            hip.check(L.ssr_usm_sharp(gt.data_ptr(), self._gt_usm.data_ptr(), self.B * self.cout, self.H, self.W, scale,
                                      0.5, 10.0, st), "ssr_usm_sharp")
            gt_usm = self._gt_usm
        for flag, buf in self._tgt_by_flag.items():      # one NHWC copy per distinct target image
            src, sc = (gt_usm, 1.0) if flag else (gt, scale)
            hip.check(L.ssr_nchw_to_nhwc(src.data_ptr(), self.B, self.cout, self.H, self.W, view(buf), self.dt, 1, 1, sc, st),
                      "target->nhwc")
        if self.cfg.feed_disc_lr:   # lr_resized = F.interpolate(lr, scale_factor=4) (nearest), :133
            for buf in (self.real_in, self.fake_in):
                hip.check(L.ssr_nchw_to_nhwc(lr.data_ptr(), self.B, self.cin, self.h, self.w,
                                             hip.View(buf.data_ptr(), buf.shape[-1], self.cout), self.dt, 1, 4, scale,
                                             st), "lr_resized")
        if old_hr is not None:      # torch.cat((.., self.old_hr), dim=1): the last 3 channels of both D inputs  (:171-175,:202-207)
            old_hr = old_hr.contiguous()
            coff = self.cout + (self.cin if self.cfg.feed_disc_lr else 0)
            for buf in (self.real_in, self.fake_in):
                hip.check(L.ssr_nchw_to_nhwc(old_hr.data_ptr(), self.B, self.cout, self.H, self.W,
                                             hip.View(buf.data_ptr(), buf.shape[-1], coff), self.dt, 1, 1, scale, st), "old_hr")

    # ------------------------------------------------------------------ phases
    @staticmethod
    def _zero(t: torch.Tensor):
        """fp32 arena <- 0 by the library's own fill kernel: no ATen launch inside the (captured) step"""
        hip.check(hip.lib().ssr_fill(t.data_ptr(), t.numel(), hip.F32, 0.0, hip.stream_ptr()), "ssr_fill")

    def _bce(self, target, weight, loss_idx, mean_idx, with_grad=True, plan=None):
        d = plan or self.d_plan
        lp, ls = self.losses.data_ptr(), 4 * self.loss_stride
        hip.check(hip.lib().ssr_bce_logits_loss(view(d.logits), view(d.d_logits) if with_grad else hip.NULL_VIEW,
                                                self.loss_dt, self.B * self.H * self.W, target, weight, lp + ls * loss_idx,
                                                (lp + ls * mean_idx) if mean_idx is not None else None,
                                                hip.stream_ptr()), "ssr_bce_logits_loss")

    def _d_forward(self, x_buf):
        self.d_store.spectral_norm(power_iter=True)   # train-mode hook: one power iteration per forward
        self.d_store.pack()
        self.d_plan.forward_plan(x_buf).run()

    def _phase_g(self, run_bwd: bool = True):
        cfg = self.cfg
        self._zero(self.g_store.grad)
        self._zero(self.losses)
        self.g_store.pack()
        self.g_plan.fwd.run()                                              # :140
        hip.check(hip.lib().ssr_l1_loss(view(self.fake_in), view(self.l1_tgt), view(self.grad_l1), self.loss_dt,
                                        self.B * self.H * self.W, self.cout, cfg.l1_weight, self.losses.data_ptr(),
                                        hip.stream_ptr()), "ssr_l1_loss")   # :147-150
        if self.p_plan is not None:                                        # :153-160
            self.p_plan.fwd_target.run()
            self.p_plan.fwd.run()
            self.p_plan.bwd.run()
        self._d_forward(self.fake_in)                                      # :181
        self._bce(cfg.real_label, cfg.gan_weight, 1, None)                 # :182 (is_disc=False)
        self.d_plan.backward_plan(self.fake_in, param_grads=False, input_grad=True,
                                  in_residual=self.grad_l1).run()          # :192, D frozen (:136-137)
        if run_bwd:
            self.g_plan.bwd.run()

    def _phase_g_skipped(self):
        """current_iter fails the gate at :144: only the forward runs (self.output is still needed)."""
        self._zero(self.losses)
        self.g_store.pack()
        self.g_plan.fwd.run()

    def _phase_d(self):
        cfg = self.cfg
        self._zero(self.d_store.grad)                                      # optimizer_d.zero_grad() :215
        self._zero(self.d_store.grad_sn)
        self._d_forward(self.real_in)                                      # :217
        self._bce(cfg.real_label, 1.0, 2, 3)                               # :218-220
        self.d_plan.backward_plan(self.real_in, param_grads=True, input_grad=False).run()   # :221
        self.d_store.spectral_norm_backward()
        self._d_forward(self.fake_in)                                      # :224 (output.detach())
        self._bce(cfg.fake_label, 1.0, 4, 5)                               # :225-226
        self.d_plan.backward_plan(self.fake_in, param_grads=True, input_grad=False).run()   # :227
        self.d_store.spectral_norm_backward()

    def _phase_opt_g(self):
        self.opt_g.update(self.dp.grad_scale)                              # :193 (+ EMA :230-231)

    def _phase_opt_d(self):
        self.opt_d.update(self.dp.grad_scale)                              # :228

    def _phase_ema_only(self):
        """model_ema runs on EVERY iteration (:230-231), also when the gate at :144 kept G's optimizer from stepping (then
        the fused Adam+EMA launch does not run): ema = ema*decay + p*(1-decay)."""
        o = self.opt_g
        if o.ema is not None:
            dec = self.cfg.ema_decay
            hip.check(hip.lib().ssr_axpby_f32(1.0 - dec, o.store.data.data_ptr(), dec, o.ema.data_ptr(), o.store.numel,
                                              hip.stream_ptr()), "ssr_axpby_f32 (ema)")

    # ------------------------------------------------------------------ driver
    def _run(self, name, fn):
        if not self.use_graph:
            fn()
            return
        g = self._graphs.get(name)
        if g is None:
            if name not in self._warm:     # first touch of every kernel of THIS launch list (hipFuncSetAttribute, lazily
                self._warm.add(name)       # built backward plans) outside capture; captured on its second use
                return fn()
            g = torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls are policed during capture.  With RCCL the process group's watchdog
            # thread queries events at any time; in the default global mode such a call from another thread invalidates
            # the capture.  Everything captured here is launched from this thread.
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
            self._graphs[name] = g
        g.replay()

    def _dp_step_body(self):
        """The forked data-parallel step as one stream program (captured whole when SSR_DP_ONE_GRAPH=1; see step()).
        hipGraph capture on ROCm 7.2 dies (SIGSEGV in hipStreamEndCapture) when the comm stream is ENTERED from two different streams
        (tools/graph_fork_probe.py: `full` crashes, `curonly` captures), so every exchange is forked from the main stream here: D's
        exchange is issued after G's slices, once the main stream has taken D's backward as a dependency - all it has left to do by
        then is G's Adam, which waits for G's exchanges anyway."""
        self._phase_g(run_bwd=False)
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self._phase_d()
            d_done = torch.cuda.Event()
            d_done.record(side)
        segs = getattr(self.g_plan, "bwd_segments", None) or [(self.g_plan.bwd, 0, self.g_store.numel)]
        hgs = []
        for L, off, n in segs:
            L.run()
            hgs.append(self.dp.all_reduce_async(self.g_store.grad[off:off + n]))
        cur.wait_event(d_done)
        hd = self.dp.all_reduce_async(self.d_store.grad)          # (forked from the main stream, like G's)
        with torch.cuda.stream(side):
            self.dp.wait(hd)
            self._phase_opt_d()
        for hg in hgs:
            self.dp.wait(hg)
        self._phase_opt_g()
        cur.wait_stream(side)
        cs = self.dp._stream()
        if cs is not None:
            cur.wait_stream(cs)        # (inside a capture every forked stream must join the capturing one)

    def step(self, current_iter: Optional[int] = None):
        """One optimize_parameters().  Order of device work (single rank): identical to the reference.
        With DP, G's gradient exchange is issued slice by slice behind the segments of G's backward (and so overlaps it and
        the D phases on the side stream), D's single exchange is issued after G's slices, and each Adam update waits for its
        own exchange only (the D phases do not read G's parameters, so the result is unchanged)."""
        self.iter = self.iter + 1 if current_iter is None else current_iter
        cfg = self.cfg
        g_on = (self.iter % cfg.net_d_iters == 0) and (self.iter > cfg.net_d_init_iters)
        if self.dp.active and g_on and self.overlap_d and self.dp_fork:
            # data parallel, forked: as in the single-process step the discriminator phases run on a side stream beside G's
            # backward; each network's gradient exchange follows its own backward and each Adam its own exchange.  Collectives are
            # ISSUED in the same program order on every rank (G's slices, then D's; they execute in that order on the one comm stream).
            if self.dp_one_graph:
                # SSR_DP_ONE_GRAPH=1 (RCCL only: its collectives are stream operations and can be captured; gloo's are host calls):
                # the whole data-parallel step - both phase chains, the slice exchanges on the comm stream, both Adam updates - as
                # ONE graph launch per step instead of seven graph launches with the collectives issued from Python between them
                # (8 ranks on 16 host cores).  Same device work, same order on every stream.
                self._run("dp_step", self._dp_step_body)
                return
            self._run("g_pre", lambda: self._phase_g(run_bwd=False))
            cur = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            side = self._side
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._run("d", self._phase_d)
            # G's backward, segment by segment: the exchange of segment k (the last layers first) runs on the comm stream while
            # segment k+1 computes; D's single exchange is ISSUED after them (the comm stream executes in issue order and must
            # not hold G's early slices behind the end of the D phases).  Same issue order on every rank.
            segs = getattr(self.g_plan, "bwd_segments", None) or [(self.g_plan.bwd, 0, self.g_store.numel)]
            hgs = []
            for k, (L, off, n) in enumerate(segs):
                self._run(f"g_bwd{k}" if len(segs) > 1 else "g_bwd", L.run)
                hgs.append(self.dp.all_reduce_async(self.g_store.grad[off:off + n]))
            with torch.cuda.stream(side):
                hd = self.dp.all_reduce_async(self.d_store.grad)
                self.dp.wait(hd)
                self._run("opt_d", self._phase_opt_d)
            for hg in hgs:
                self.dp.wait(hg)
            self._run("opt_g", self._phase_opt_g)
            cur.wait_stream(side)
        elif self.dp.active:
            hg = None
            if g_on:
                self._run("g", self._phase_g)
                hg = self.dp.all_reduce_async(self.g_store.grad)
            else:
                self._run("g_skip", self._phase_g_skipped)
            self._run("d", self._phase_d)
            hd = self.dp.all_reduce_async(self.d_store.grad)
            if g_on:
                self.dp.wait(hg)                                   # G's Adam runs under D's exchange
                self._run("opt_g", self._phase_opt_g)
            else:
                self._run("ema", self._phase_ema_only)
            self.dp.wait(hd)
            self._run("opt_d", self._phase_opt_d)
        else:
            def whole():
                if not self.overlap_d:
                    self._phase_g()
                    self._phase_opt_g()
                    self._phase_d()
                    self._phase_opt_d()
                    return
                # Once D has handed its input gradient to the generator, nothing in G's backward / Adam / EMA depends on the
                # discriminator phases and vice versa (they read G's OUTPUT of this iteration, `self.output.detach()`, :224, and
                # D's own parameters): the two run on two streams — a fork / join inside the captured graph — so that the
                # workgroups of one fill the launch ramps and tails of the other (one dependent launch chain alone keeps
                # the CUs busy 78 % of the time: profiles/r02b_pmc_sq.json).  Same arithmetic, same results.
                self._phase_g(run_bwd=False)
                cur = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream()
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self._phase_d()
                    self._phase_opt_d()
                self.g_plan.bwd.run()
                self._phase_opt_g()
                cur.wait_stream(self._side)

            def whole_skip():
                self._phase_g_skipped()
                self._phase_d()
                self._phase_opt_d()
                self._phase_ema_only()
            self._run("step" if g_on else "step_skip", whole if g_on else whole_skip)

    # ------------------------------------------------------------------ results
    def log(self) -> "OrderedDict[str, float]":
        """get_current_log(): one host sync, only when the caller logs (train.py:116-121)."""
        scal = self.losses if not self.det else self.losses.view(8, self.loss_stride).double().cpu().sum(1).float().to(self.losses.device)
        vals = self.dp.reduce_scalars(scal).tolist()      # det: the per-block slots added in index order on the host
        out = OrderedDict((k, vals[i]) for i, k in enumerate(LOSS_KEYS))
        if self.p_plan is not None:
            out["l_g_percep"] = vals[6]
        return out

    def output(self) -> torch.Tensor:
        """self.output (NCHW fp32) of the last generator forward."""
        return self.g_plan.read_output()
