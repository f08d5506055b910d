"""Shared plumbing of the arch plugins: an nn.Module whose parameters carry the reference's
state_dict names/shapes (so published checkpoints load with strict=True) but whose storage is a flat
arena consumed by the HIP launch plans (engine.py).  torch.autograd sees ONE Function per network.

There is deliberately no CPU / eager implementation here: forward() on a non-GPU tensor, or without
libssr_hip.so, raises."""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine, hip


class ConvParams(nn.Module):
    """Parameter holder of one convolution (never called)."""

    def __init__(self, spec: engine.ConvSpec, init: str):
        super().__init__()
        w = torch.empty(spec.cout, spec.cin, spec.k, spec.k)
        fan_in = spec.cin * spec.k * spec.k
        bound = 1.0 / math.sqrt(fan_in)
        if init == "rdb":     # default_init_weights(..., 0.1): kaiming_normal * 0.1, zero bias (arch_util.py:600-628)
            nn.init.kaiming_normal_(w)
            w.mul_(0.1)
        else:                 # torch.nn.Conv2d default
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        if spec.sn:           # torch.nn.utils.spectral_norm naming (discriminator_arch.py:30-39)
            self.weight_orig = nn.Parameter(w)
            self.register_buffer("weight_u", F.normalize(torch.randn(spec.cout), dim=0, eps=1e-12))
            self.register_buffer("weight_v", F.normalize(torch.randn(fan_in), dim=0, eps=1e-12))
        else:
            self.weight = nn.Parameter(w)
        if spec.bias:
            b = torch.zeros(spec.cout) if init == "rdb" else torch.empty(spec.cout).uniform_(-bound, bound)
            self.bias = nn.Parameter(b)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("ConvParams only holds parameters; the network runs through the HIP plan")


def attach(root: nn.Module, dotted: str, leaf: nn.Module):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, nn.Module())
        m = getattr(m, p)
    m.add_module(parts[-1], leaf)


class HipNet(nn.Module):
    def __init__(self, specs: List[engine.ConvSpec], compute_dtype="fp32", rdb_prefix: str = "body."):
        super().__init__()
        self._specs = specs
        self.compute_dtype = compute_dtype
        for s in specs:
            attach(self, s.name, ConvParams(s, "rdb" if s.name.startswith(rdb_prefix) else "default"))
        self._store = None
        self._plans: Dict[Tuple, object] = {}
        self._packed_version = None
        self._frozen = False

    # ---- flat-arena storage ----
    def _leaf(self, name: str) -> nn.Module:
        m = self
        for p in name.split("."):
            m = getattr(m, p)
        return m

    def param_keys(self) -> List[str]:
        return [k for k, _ in self.named_parameters()]

    def store(self) -> engine.ParamStore:
        """(Re)build the arena on the parameters' device and alias every parameter/buffer into it."""
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise hip.HipLibraryError("this network runs only on the GPU through libssr_hip.so (no CPU path): "
                                      "move it with .to('cuda') / .cuda()")
        st = self._store
        first_key = getattr(self, "_first_key", None)        # the module tree is static: one traversal, not one per forward
        if first_key is None:
            first_key = self._first_key = self.param_keys()[0]
        if st is not None and st.device == p0.device and p0.data_ptr() == st.ptr(first_key):
            return st
        hip.lib()
        st = engine.ParamStore(self._specs, hip.dtype_code(self.compute_dtype), device=p0.device)
        with torch.no_grad():
            for key, p in self.named_parameters():
                st.tensor(key).copy_(p.data)
                p.data = st.tensor(key)
            for s in self._specs:
                if s.sn:
                    leaf = self._leaf(s.name)
                    st.u[s.name].copy_(leaf.weight_u)
                    st.v[s.name].copy_(leaf.weight_v)
                    leaf.weight_u = st.u[s.name]
                    leaf.weight_v = st.v[s.name]
        self._store = st
        self._plans.clear()
        self._packed_version = None
        return st

    def pack_if_stale(self):
        """The kernels read packed compute-dtype copies of the weights.  By default they are re-packed on EVERY forward: a
        Parameter's version counter does not move under writes through `.data` (BasicSR's model_ema updates net_g_ema exactly that
        way on every iteration, and any user `p.data.copy_` does too), so "unchanged since the last packing" cannot be decided from
        the versions.  Inference loops that call the module once per chunk with fixed weights (infer.py / infer_grid.py) opt in to
        packing once with freeze_packed(); load_state_dict / store rebuilds / mark_weights_dirty() thaw it again."""
        st = self.store()
        if self._packed_version is None or not self._frozen:
            st.pack()
            self._packed_version = 1

    def run_forward(self, plan):
        """plan.fwd: the launch list, or - inference with frozen weights (freeze_packed) - its hipGraph: one graph launch instead
        of ~160 ctypes calls per forward (5 ms of driver-thread time per 64-chunk batch in whole-tile inference, r04k profile).
        First use of a plan runs eagerly (lazy kernel attributes), the second is captured, later ones replay."""
        import os
        if not (self._frozen and not plan.training and os.environ.get("SSR_INFER_GRAPH", "1") == "1"):
            plan.fwd.run()
            return
        g = getattr(plan, "_fwd_graph", None)
        if g is None:
            plan.fwd.run()
            plan._fwd_graph = "warm"
        elif g == "warm":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                plan.fwd.run()
            plan._fwd_graph = g
            g.replay()
        else:
            g.replay()

    def freeze_packed(self, on: bool = True):
        """Explicit promise that the parameters will not change until freeze_packed(False): pack now, skip packing afterwards."""
        self._frozen = bool(on)
        self._packed_version = None
        if on:
            self.pack_if_stale()
        return self

    def mark_weights_dirty(self):
        self._packed_version = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._frozen = False            # new weights: thaw (the caller freezes again if it wants to)
        self._packed_version = None
        return out

    def grads_from_arena(self, needs: List[bool]):
        st = self._store
        return [st.tensor(k, st.grad).clone() if need else None for k, need in zip(self.param_keys(), needs)]
