"""SSR_RRDBNet on MI355X — same registry name, ctor kwargs, forward contract and state_dict layout as
/root/reference/ssr/archs/rrdbnet_arch.py:71-137 (generator: conv_first widened to n_frames*bands input
channels, `num_block` RRDBs of 3 dense blocks, two nearest-x2 + conv stages, conv_hr, conv_last).

forward(x: float32[B, num_in_ch, H, W]) -> float32[B, num_out_ch, 4H, 4W]  (H, W arbitrary).
The work is done by engine.GeneratorPlan through libssr_hip.so; torch.autograd sees one Function."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import engine, hip
from ..registry import ARCH_REGISTRY
from .hipnet import HipNet


def _generator_forward(net, need, x):
    net.store()
    B, _, H, W = x.shape
    plan = net.plan(B, H, W, training=need)
    net.pack_if_stale()
    plan.load_input(x.detach().contiguous().float())
    net.run_forward(plan)
    plan.generation = getattr(plan, "generation", 0) + 1
    return plan, plan.read_output()


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, need, x, *params):
        plan, y = _generator_forward(net, need, x)
        ctx.net, ctx.plan, ctx.gen = net, plan, plan.generation
        return y

    @staticmethod
    def backward(ctx, gy):
        net, plan = ctx.net, ctx.plan
        if not plan.training or plan.generation != ctx.gen:
            raise RuntimeError("SSR_RRDBNet: activations of this forward were overwritten by a later forward of "
                               "the same shape; call backward() before the next forward (as the reference loop does)")
        st = net.store()
        plan.load_output_grad(gy.contiguous().float())
        st.grad.zero_()
        plan.bwd.run()
        gx = None
        if ctx.needs_input_grad[2]:
            gx = plan.read_input_grad()
            if plan.unshuffle > 1:
                gx = F.pixel_shuffle(gx, plan.unshuffle)   # inverse of the pixel_unshuffle index map
        return (None, None, gx, *net.grads_from_arena(list(ctx.needs_input_grad[3:])))


@ARCH_REGISTRY.register()
class SSR_RRDBNet(HipNet):
    def __init__(self, num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                 compute_dtype="fp32"):
        self.kwargs = dict(num_in_ch=num_in_ch, num_out_ch=num_out_ch, scale=scale, num_feat=num_feat,
                           num_block=num_block, num_grow_ch=num_grow_ch)
        super().__init__(engine.generator_specs(**self.kwargs), compute_dtype)
        self.scale = scale

    def plan(self, B, H, W, training):
        key = (B, H, W, bool(training))
        if key not in self._plans:
            self._plans[key] = engine.GeneratorPlan(self._store, B, H, W, training=training,
                                                    need_input_grad=training, **self.kwargs)
        return self._plans[key]

    def forward(self, x):
        if not torch.is_grad_enabled():      # inference: no autograd node (and no marshalling of 702 parameters through Function.apply)
            return _generator_forward(self, False, x)[1]
        params = list(self.parameters())
        # grad mode is off inside Function.forward: decide here whether activations must be retained
        need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        return _GeneratorFn.apply(self, need, x, *params)
