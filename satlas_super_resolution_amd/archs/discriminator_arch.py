"""SSR_UNetDiscriminatorSN on MI355X — same registry name, ctor kwargs, forward contract and state_dict
layout (conv0/conv9 weight+bias; conv1..8 weight_orig + buffers weight_u / weight_v) as
/root/reference/ssr/archs/discriminator_arch.py:11-71.

forward(x: float32[B, num_in_ch, H, W]) -> float32[B, 1, H, W] logits (H, W divisible by 8).  In
train() mode every forward performs one spectral-norm power iteration in place on weight_u/weight_v,
exactly like the hook-style torch.nn.utils.spectral_norm the reference wraps its convs with."""
from __future__ import annotations

import torch

from .. import engine, hip
from ..registry import ARCH_REGISTRY
from .hipnet import HipNet


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        st = net.store()
        B, C, H, W = x.shape
        plan, xin = net.plan(B, H, W)
        st.spectral_norm(power_iter=net.training)
        st.pack()
        xc = x.detach().contiguous().float()
        hip.check(hip.lib().ssr_nchw_to_nhwc(xc.data_ptr(), B, C, H, W, hip.view(xin), st.dtype, 1, 1, 1.0,
                                             hip.stream_ptr()), "ssr_nchw_to_nhwc")
        plan.forward_plan(xin).run()
        plan.generation = getattr(plan, "generation", 0) + 1
        ctx.net, ctx.plan, ctx.xin, ctx.gen = net, plan, xin, plan.generation
        out = torch.empty(B, 1, H, W, device=x.device)
        hip.check(hip.lib().ssr_nhwc_to_nchw(hip.view(plan.logits), st.dtype, out.data_ptr(), B, 1, H, W,
                                             hip.stream_ptr()), "ssr_nhwc_to_nchw")
        return out

    @staticmethod
    def backward(ctx, gy):
        net, plan, xin = ctx.net, ctx.plan, ctx.xin
        if plan.generation != ctx.gen:
            raise RuntimeError("SSR_UNetDiscriminatorSN: activations of this forward were overwritten by a later "
                               "forward; call backward() before the next forward (as the reference loop does)")
        st = net.store()
        B, _, H, W = gy.shape
        g = gy.contiguous().float()
        hip.check(hip.lib().ssr_nchw_to_nhwc(g.data_ptr(), B, 1, H, W, hip.view(plan.d_logits), st.dtype, 1, 1, 1.0,
                                             hip.stream_ptr()), "ssr_nchw_to_nhwc")
        needs = list(ctx.needs_input_grad[2:])
        pg, ig = any(needs), ctx.needs_input_grad[1]
        st.grad.zero_()
        st.grad_sn.zero_()
        plan.backward_plan(xin, param_grads=pg, input_grad=ig).run()
        if pg:
            st.spectral_norm_backward()
        gx = None
        if ig:
            gx = torch.empty(B, net.num_in_ch, H, W, device=gy.device)
            hip.check(hip.lib().ssr_nhwc_to_nchw(hip.view(plan.g_in), st.dtype, gx.data_ptr(), B, net.num_in_ch, H, W,
                                                 hip.stream_ptr()), "ssr_nhwc_to_nchw")
        return (None, gx, *net.grads_from_arena(needs))


@ARCH_REGISTRY.register()
class SSR_UNetDiscriminatorSN(HipNet):
    def __init__(self, num_in_ch, num_feat=64, skip_connection=True, compute_dtype="fp32"):
        super().__init__(engine.discriminator_specs(num_in_ch, num_feat), compute_dtype, rdb_prefix="\0")
        self.num_in_ch, self.num_feat, self.skip_connection = num_in_ch, num_feat, skip_connection

    def plan(self, B, H, W):
        key = (B, H, W)
        if key not in self._plans:
            p = engine.DiscriminatorPlan(self._store, B, H, W, num_in_ch=self.num_in_ch, num_feat=self.num_feat,
                                         skip_connection=self.skip_connection)
            xin = torch.zeros(B, H, W, p.cdp, dtype=hip.torch_dtype(self._store.dtype), device=self._store.device)
            self._plans[key] = (p, xin)
        return self._plans[key]

    def forward(self, x):
        return _DiscriminatorFn.apply(self, x, *self.parameters())
