"""Launch plans for the ESRGAN hot path on one MI355X.

A *plan* is a static list of C-ABI calls (include/ssr_hip.h) over preallocated NHWC device buffers:
  GeneratorPlan      SSR_RRDBNet forward / backward        (/root/reference/ssr/archs/rrdbnet_arch.py:116-137)
  DiscriminatorPlan  SSR_UNetDiscriminatorSN fwd / bwd     (/root/reference/ssr/archs/discriminator_arch.py:42-71)
Because shapes are static, a whole plan (or the whole train step built from plans) can be captured in
one hipGraph.  Dense blocks are concat-free: every RDB owns one [B,H,W,nf+4*gc] buffer, conv_k reads the
channel prefix and writes its own 32-channel slice; the matching gradient buffer receives dgrad
fan-in by in-place accumulation with the LeakyReLU-backward mask applied by the *last* contributor.
All weight gradients of the 3x3 layers are computed by ONE batched wgrad launch at the end.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import hip
from .hip import ConvDesc, View, WgradItem, WgradLayer, PackItem, SNItem, SNBwdItem, ReduceItem, view


def rup(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def ck_of(dt: int) -> int:
    return 32 if dt == hip.BF16 else 16


@dataclass
class ConvSpec:
    name: str
    cout: int
    cin: int
    k: int = 3
    stride: int = 1
    bias: bool = True
    sn: bool = False  # spectral-normalised (weight stored as weight_orig)
    dgrad_packed: bool = True  # keep a rotated/transposed copy for the per-conv (scatter) dgrad
    s2d: Optional[bool] = None  # 4x4 stride-2 layers: None = space-to-depth path when the shape allows, False = never


class ParamStore:
    """Flat fp32 arenas (param, grad) for one network in the reference's state_dict layout, plus the
    packed compute-dtype weight copies the kernels read.

    Reference key layout: `<conv>.weight`/`.bias`, or `.weight_orig` (+ buffers
    `.weight_u`, `.weight_v`) for spectral-normalised layers."""

    def __init__(self, specs: List[ConvSpec], dtype: int, device="cuda"):
        self.specs = OrderedDict((s.name, s) for s in specs)
        # modes hip.F32F / hip.F32H: forward convolutions exact fp32 / fp16-split, everything else (storage, backward convolutions, weight gradients) as fp32x3;
        # `dtype` is what the C ABI sees for tensors and backward launches, `fwd_dtype` what forward conv descriptors and the forward
        # packing carry
        self.mode = dtype
        self.fwd_dtype = hip.forward_code(dtype)
        dtype = hip.storage_code(dtype)
        self.dtype = dtype
        self.device = torch.device(device)
        self.offsets: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        off = 0
        for s in specs:
            wname = s.name + (".weight_orig" if s.sn else ".weight")
            shape = (s.cout, s.cin, s.k, s.k)
            self.offsets[wname] = (off, shape)
            off += s.cout * s.cin * s.k * s.k
            off = rup(off, 4)  # 16-byte aligned tensors
            if s.bias:
                self.offsets[s.name + ".bias"] = (off, (s.cout,))
                off += rup(s.cout, 4)
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        # spectral-norm state
        self.sn_names = [s.name for s in specs if s.sn]
        self.u: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        for s in specs:
            if s.sn:
                self.u[s.name] = torch.zeros(s.cout, device=self.device)
                self.v[s.name] = torch.zeros(s.cin * s.k * s.k, device=self.device)
        if self.sn_names:
            self.sigma = torch.ones(len(self.sn_names), device=self.device)
            mx = max(self.specs[n].cout + self.specs[n].cin * self.specs[n].k ** 2 for n in self.sn_names)
            self.sn_tmp = torch.zeros(len(self.sn_names), rup(mx + 4, 4), device=self.device)
            # dW w.r.t. the normalised weight (wgrad target for SN layers), same offsets as `grad`
            self.grad_sn = torch.zeros(off, dtype=torch.float32, device=self.device)
            self.sn_dot = torch.zeros(len(self.sn_names), hip.SN_BWD_SLOTS, device=self.device)   # per-block partial sums of <dW_sn, W> (fixed-order sum in the kernel)
        # packed weights
        tdt = hip.torch_dtype(dtype)
        L = hip.lib()
        self.packed_fwd: Dict[str, torch.Tensor] = {}
        self.packed_dgrad: Dict[str, torch.Tensor] = {}
        self.pad: Dict[str, Tuple[int, int, int, int]] = {}
        self.s2d: Dict[str, bool] = {}
        items = []
        fdt = self.fwd_dtype
        for i, s in enumerate(specs):
            kk = s.k * s.k
            ck_f = L.ssr_conv2d_ck(fdt, s.k)                         # kernel consuming the forward weights
            ck_d = L.ssr_conv2d_ck(dtype, s.k if s.stride == 1 else 2)   # dgrad kernel (2x2 parity classes for s2)
            # 4x4 stride-2 layers (discriminator_arch.py:31-35) run as 2x2 layers over a space-to-depth view of their
            # input when the shape allows (ssr_conv_desc.s2d): the forward weights are packed in that order
            s2d = bool(s.k == 4 and s.stride == 2 and s.s2d is not False and L.ssr_conv2d_s2d_ok(fdt, s.cin, s.cout, rup(s.cout, 32)))
            self.s2d[s.name] = s2d
            if s2d:
                ck_f = 32 if fdt == hip.BF16 else 16       # (split-bf16 mode: 16-channel chunks, csrc/conv_big_x3.hip)
            cout_pad, cin_pad = rup(s.cout, 32), rup(rup(s.cin, 8), ck_f)
            cin_pad_o, cout_pad_i = rup(s.cin, 32), rup(rup(s.cout, 8), ck_d)
            self.pad[s.name] = (cout_pad, cin_pad, cin_pad_o, cout_pad_i)
            pf = torch.zeros(kk * cout_pad * cin_pad, dtype=tdt, device=self.device)
            ntap_d = kk if s.stride == 1 else 16
            pd = torch.zeros(ntap_d * cin_pad_o * cout_pad_i if s.dgrad_packed else 8, dtype=tdt, device=self.device)
            self.packed_fwd[s.name], self.packed_dgrad[s.name] = pf, pd
            woff, _ = self.offsets[s.name + (".weight_orig" if s.sn else ".weight")]
            inv = (self.sigma.data_ptr() + 4 * self.sn_names.index(s.name)) if s.sn else None
            items.append(PackItem(self.data.data_ptr() + 4 * woff, inv, pf.data_ptr(),
                                  pd.data_ptr() if s.dgrad_packed else None,
                                  s.cout, s.cin, s.k, s.k, s.stride, cout_pad, cin_pad, cin_pad_o, cout_pad_i, ck_f, ck_d,
                                  1 if s2d else 0))
        self._pack_items = items
        # mixed mode: the forward packing in the forward kernels' layout (one launch), the backward packing in the backward kernels' (a second)
        self._pack_items_bwd = None
        if fdt != dtype:
            fw, bw = [], []
            for it in items:
                a, b = PackItem(), PackItem()
                C.memmove(C.byref(a), C.byref(it), C.sizeof(PackItem))
                C.memmove(C.byref(b), C.byref(it), C.sizeof(PackItem))
                a.dst_dgrad = None
                b.dst_fwd = None
                fw.append(a)
                bw.append(b)
            self._pack_items, self._pack_items_bwd = fw, bw
            items = fw
            self.pack_table_bwd = hip.device_table(bw)
        self.repacked: Dict[Tuple[str, int], torch.Tensor] = {}
        # gathered dense-block dgrad weights (filled by add_rdb_gather)
        self.gather: Dict[Tuple[str, int], torch.Tensor] = {}
        self.gather_rows: Dict[Tuple[str, int], int] = {}
        self._seg_items: List[hip.PackSeg] = []
        self.seg_table = None
        self.pack_table = hip.device_table(items)
        if self.sn_names:
            sn_items, bwd_items = [], []
            for j, n in enumerate(self.sn_names):
                s = self.specs[n]
                woff, _ = self.offsets[n + ".weight_orig"]
                rows, cols = s.cout, s.cin * s.k * s.k
                sn_items.append(SNItem(self.data.data_ptr() + 4 * woff, self.u[n].data_ptr(), self.v[n].data_ptr(),
                                       self.sigma.data_ptr() + 4 * j, self.sn_tmp[j].data_ptr(), rows, cols))
                bwd_items.append(SNBwdItem(self.grad_sn.data_ptr() + 4 * woff, self.data.data_ptr() + 4 * woff,
                                           self.u[n].data_ptr(), self.v[n].data_ptr(), self.sigma.data_ptr() + 4 * j,
                                           self.grad.data_ptr() + 4 * woff, self.sn_dot[j].data_ptr(), rows, cols))
            self.sn_table = hip.device_table(sn_items)
            self.sn_bwd_table = hip.device_table(bwd_items)
            self.sn_max_rows = max(self.specs[n].cout for n in self.sn_names)
            self.sn_max_cols = max(self.specs[n].cin * self.specs[n].k ** 2 for n in self.sn_names)
            self.sn_max_elems = max(self.specs[n].cout * self.specs[n].cin * self.specs[n].k ** 2
                                    for n in self.sn_names)

    # ---- tensor views in the reference layout ----
    def tensor(self, key: str, arena: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, shape = self.offsets[key]
        n = math.prod(shape)
        return (self.data if arena is None else arena)[off:off + n].view(shape)

    def ptr(self, key: str, arena: Optional[torch.Tensor] = None) -> int:
        off, _ = self.offsets[key]
        return (self.data if arena is None else arena).data_ptr() + 4 * off

    def wkey(self, name: str) -> str:
        return name + (".weight_orig" if self.specs[name].sn else ".weight")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        with torch.no_grad():
            for key in self.offsets:
                if key in sd:
                    self.tensor(key).copy_(sd[key].to(self.device, torch.float32))
                elif strict:
                    raise KeyError(key)
            for n in self.sn_names:
                if n + ".weight_u" in sd:
                    self.u[n].copy_(sd[n + ".weight_u"].to(self.device, torch.float32))
                    self.v[n].copy_(sd[n + ".weight_v"].to(self.device, torch.float32))
                elif strict:
                    raise KeyError(n + ".weight_u")

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        out = OrderedDict()
        for s in self.specs.values():
            if s.sn:
                out[s.name + ".weight_orig"] = self.tensor(s.name + ".weight_orig").clone()
                out[s.name + ".weight_u"] = self.u[s.name].clone()
                out[s.name + ".weight_v"] = self.v[s.name].clone()
            else:
                out[s.name + ".weight"] = self.tensor(s.name + ".weight").clone()
            if s.bias:
                out[s.name + ".bias"] = self.tensor(s.name + ".bias").clone()
        return out

    # ---- device ops ----
    def pack(self):
        hip.check(hip.lib().ssr_pack_weights(self.pack_table.data_ptr(), len(self._pack_items), self.fwd_dtype,
                                             hip.stream_ptr()), "ssr_pack_weights")
        if self._pack_items_bwd is not None:
            hip.check(hip.lib().ssr_pack_weights(self.pack_table_bwd.data_ptr(), len(self._pack_items_bwd), self.dtype,
                                                 hip.stream_ptr()), "ssr_pack_weights (backward layouts)")
        if self._seg_items:
            if self.seg_table is None:
                self.seg_table = hip.device_table(self._seg_items)
            hip.check(hip.lib().ssr_pack_dgrad_gather(self.seg_table.data_ptr(), len(self._seg_items), self.dtype,
                                                      hip.stream_ptr()), "ssr_pack_dgrad_gather")

    def add_repack(self, name: str, ck: int) -> torch.Tensor:
        """An extra forward packing [K chunk of `ck`][tap][CoutPad][ck] of layer `name` (refreshed by pack() like the
        default one).  The fused dense-block kernel streams conv5 in 16-channel chunks (csrc/rdb_fwd.hip)."""
        key = (name, ck)
        if key in self.repacked:
            return self.repacked[key]
        s = self.specs[name]
        cout_pad, cin_pad = rup(s.cout, 32), rup(rup(s.cin, 8), ck)
        buf = torch.zeros(s.k * s.k * cout_pad * cin_pad, dtype=hip.torch_dtype(self.dtype), device=self.device)
        woff, _ = self.offsets[name + (".weight_orig" if s.sn else ".weight")]
        inv = (self.sigma.data_ptr() + 4 * self.sn_names.index(name)) if s.sn else None
        self._pack_items.append(PackItem(self.data.data_ptr() + 4 * woff, inv, buf.data_ptr(), None, s.cout, s.cin, s.k, s.k,
                                         s.stride, cout_pad, cin_pad, rup(s.cin, 32), rup(rup(s.cout, 8), ck), ck, ck))
        self.pack_table = hip.device_table(self._pack_items)
        self.repacked[key] = buf
        return buf

    def add_rdb_gather(self, prefix: str, nf: int, gc: int, a5: float, ck0: int = 0):
        """Packed weights of the gather-form backward of one ResidualDenseBlock (rrdbnet_arch.py:37-44):
        slice k (0: the 64-ch block input, 1..4: x1..x4) <- conv3x3 over [dpre_{k+1}..dpre_4 | d_out],
        conv5's part pre-scaled by a5 (0.2, or 0.04 inside the third RDB: :44,:68).  ck0 > 0 adds a second packing of
        slice 0 in K chunks of ck0 (key (prefix, 'k0', ck0)) for the fused backward kernel."""
        tdt = hip.torch_dtype(self.dtype)
        if ck0 and (prefix, "k0", ck0) not in self.gather:
            K = 4 * gc + nf
            rows_pad, kpad = rup(nf, 32), rup(K, ck0)
            buf = torch.zeros(kpad * 9 * rows_pad, dtype=tdt, device=self.device)
            self.gather[(prefix, "k0", ck0)] = buf
            for jj in range(1, 6):
                cout_j = nf if jj == 5 else gc
                self._seg_items.append(hip.PackSeg(self.ptr(f"{prefix}.conv{jj}.weight"), buf.data_ptr(),
                                                   a5 if jj == 5 else 1.0, cout_j, nf + (jj - 1) * gc, 0, nf,
                                                   (jj - 1) * gc, rows_pad, ck0))
            self.seg_table = None
        if (prefix, 0) in self.gather:
            return
        ck = hip.lib().ssr_conv2d_ck(self.dtype, 3)
        for k in range(5):
            nout = nf if k == 0 else gc
            ci0 = 0 if k == 0 else nf + (k - 1) * gc
            K = (4 - k) * gc + nf
            rows_pad, kpad = rup(nout, 32), rup(K, ck)
            buf = torch.zeros(kpad * 9 * rows_pad, dtype=tdt, device=self.device)
            self.gather[(prefix, k)], self.gather_rows[(prefix, k)] = buf, rows_pad
            for jj in range(k + 1, 6):
                cout_j = nf if jj == 5 else gc
                cin_j = nf + (jj - 1) * gc
                self._seg_items.append(hip.PackSeg(self.ptr(f"{prefix}.conv{jj}.weight"), buf.data_ptr(),
                                                   a5 if jj == 5 else 1.0, cout_j, cin_j, ci0, nout,
                                                   (jj - k - 1) * gc, rows_pad, ck))
        self.seg_table = None

    def spectral_norm(self, power_iter: bool):
        if self.sn_names:
            hip.check(hip.lib().ssr_spectral_norm(self.sn_table.data_ptr(), len(self.sn_names), self.sn_max_rows,
                                                  self.sn_max_cols, 1 if power_iter else 0, hip.stream_ptr()),
                      "ssr_spectral_norm")

    def spectral_norm_backward(self):
        """grad += d(W/sigma)^T grad_sn ; consumes (and re-zeroes) grad_sn."""
        if self.sn_names:
            hip.check(hip.lib().ssr_spectral_norm_bwd(self.sn_bwd_table.data_ptr(), len(self.sn_names),
                                                      self.sn_max_elems, hip.stream_ptr()), "ssr_spectral_norm_bwd")
            hip.check(hip.lib().ssr_fill(self.grad_sn.data_ptr(), self.grad_sn.numel(), hip.F32, 0.0, hip.stream_ptr()), "ssr_fill")


class Launcher:
    """An ordered list of prepared C-ABI calls.  `fork(sub)` runs another launcher on a side stream from this point on (the side
    stream first waits for everything issued so far; capturable: a fork / join inside a hipGraph), `join()` makes the main stream
    wait for the side stream."""

    FORK, JOIN = object(), object()

    def __init__(self):
        self.calls = []
        self._sides = {}          # fork index -> its own side stream (forks of one launcher run concurrently)

    def add(self, fn, *args, what=""):
        self.calls.append((fn, args, what))

    def fork(self, sub: "Launcher", what=""):
        self.calls.append((Launcher.FORK, (sub,), what))

    def join(self):
        self.calls.append((Launcher.JOIN, (), "join"))

    def flat_calls(self):
        """every C-ABI call in issue order, forks inlined (instrumentation)"""
        for fn, args, what in self.calls:
            if fn is Launcher.FORK:
                yield from args[0].flat_calls()
            elif fn is not Launcher.JOIN:
                yield fn, args, what

    def run(self):
        st = hip.stream_ptr()
        for fn, args, what in self.calls:
            if fn is Launcher.FORK:
                side = self._sides.get(id(args[0]))
                if side is None:
                    side = self._sides[id(args[0])] = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    args[0].run()
                continue
            if fn is Launcher.JOIN:
                for side in self._sides.values():
                    torch.cuda.current_stream().wait_stream(side)
                continue
            rc = fn(*args, st)
            if rc != 0:
                hip.check(rc, what or getattr(fn, "__name__", "call"))

    def __len__(self):
        return len(self.calls)


class _ConvBuilder:
    """Fills ssr_conv_desc records; keeps them alive."""

    def __init__(self, store: ParamStore, N: int):
        self.store, self.N, self.dt = store, N, store.dtype
        self.keep = []
        self.packed_fwd, self.packed_dgrad = store.packed_fwd, store.packed_dgrad

    def conv(self, L: Launcher, name: str, x: View, hi: int, wi: int, y: View, *, up=1, act=hip.ACT_NONE, alpha=1.0,
             y0: View = hip.NULL_VIEW, r1: View = hip.NULL_VIEW, r1_nc=0, beta1=0.0, r2: View = hip.NULL_VIEW,
             r2_nc=0, beta2=0.0, cin: Optional[int] = None):
        s = self.store.specs[name]
        cout_pad, _, _, _ = self.store.pad[name]
        d = ConvDesc()
        d.dtype = self.store.fwd_dtype                          # (mode fp32f: forward convolutions exact, hip.F32F)
        d.x, d.N, d.Hi, d.Wi, d.up = x, self.N, hi, wi, up
        d.Cin = rup(s.cin, 8) if cin is None else cin
        d.w = self.packed_fwd[name].data_ptr()
        d.CoutPad = cout_pad
        d.bias = self.store.ptr(name + ".bias") if s.bias else None
        d.KH = d.KW = s.k
        d.stride = s.stride
        d.pad_y = d.pad_x = 1
        lh, lw = hi * up, wi * up
        d.Gh, d.Gw = (lh // s.stride, lw // s.stride)
        d.Ho, d.Wo, d.oys, d.oyo, d.oxs, d.oxo = d.Gh, d.Gw, 1, 0, 1, 0
        d.Cout = s.cout
        d.y, d.y0, d.y1 = y, y0, hip.NULL_VIEW
        d.alpha, d.act = alpha, act
        d.r1, d.r1_nc, d.beta1 = r1, r1_nc, beta1
        d.r2, d.r2_nc, d.beta2 = r2, r2_nc, beta2
        d.accumulate = 0
        d.m, d.m_c0, d.m_c1 = hip.NULL_VIEW, 0, 0
        d.s2d = 1 if self.store.s2d.get(name) else 0
        self.keep.append(d)
        fix = (self.store.fwd_dtype == hip.F32X3 and act == hip.ACT_LRELU and s.stride == 1 and s.k == 3 and alpha == 1.0 and not y0.p
               and not r1.p and not r2.p and X3_FIXUP[0])
        if fix:
            # split-bf16 mode: the LeakyReLU decisions of this layer are those of an exact evaluation - outputs whose pre-activation
            # is at rounding level (|v| < X3_FIX_THR) are listed by the epilogue and recomputed in double from the fp32 inputs and
            # the reference-layout fp32 weights by the launch that follows (include/ssr_hip.h, ssr_conv_desc.fix_*)
            d.fix_list = self._fix_list()
            d.fix_cap, d.fix_thr = X3_FIX_CAP, X3_FIX_THR[0]
            d.w_ref, d.w_ref_cin = self.store.ptr(name + (".weight_orig" if s.sn else ".weight")), s.cin
            d.w_ref_sigma = (self.store.sigma.data_ptr() + 4 * self.store.sn_names.index(name)) if s.sn else None
        L.add(hip.lib().ssr_conv2d, C.byref(d), what=f"conv fwd {name}")
        if fix:
            L.add(hip.lib().ssr_conv2d_fixup, C.byref(d), what=f"lrelu decision fix-up {name}")
        return d

    def _fix_list(self) -> int:
        """device address of a zeroed fix-up list (4 header words + X3_FIX_CAP (pixel, channel) pairs) of its own"""
        words = 4 + 2 * X3_FIX_CAP
        if not getattr(self, "_fix_pool", None) or self._fix_used == self._fix_pool[-1].shape[0]:
            self._fix_pool = getattr(self, "_fix_pool", []) + [torch.zeros(64, words, dtype=torch.int32, device=self.store.device)]
            self._fix_used = 0
        self._fix_used += 1
        return self._fix_pool[-1][self._fix_used - 1].data_ptr()

    def fix_high_water(self) -> int:
        """largest number of outputs any fix-up list of this builder has held (diagnostics / tests: must stay below X3_FIX_CAP)"""
        return max([int(p[:, 2].max()) for p in getattr(self, "_fix_pool", [])] or [0])

    def dgrad(self, L: Launcher, name: str, dy: View, gh: int, gw: int, y: View, *, cout: Optional[int] = None,
              alpha=1.0, y1: View = hip.NULL_VIEW, r1: View = hip.NULL_VIEW, r1_nc=0, beta1=0.0,
              r2: View = hip.NULL_VIEW, r2_nc=0, beta2=0.0, accumulate=0, m: View = hip.NULL_VIEW, m_c0=0, m_c1=0,
              cin_dy: Optional[int] = None, m_relu=0):
        """Gradient w.r.t. the conv input.  `dy` lives on the forward output grid (gh x gw).
        stride 1: one 3x3 conv with rotated weights; stride 2 (4x4): four 2x2 parity-class launches."""
        s = self.store.specs[name]
        _, _, cin_pad_o, cout_pad_i = self.store.pad[name]
        wbase = self.packed_dgrad[name].data_ptr()
        esz = 2 if self.dt == hip.BF16 else 4
        classes = [(0, 0)] if s.stride == 1 else [(0, 0), (0, 1), (1, 0), (1, 1)]
        arr = (ConvDesc * len(classes))()
        for ic, (py, px) in enumerate(classes):
            d = arr[ic]
            d.dtype = self.dt
            d.x, d.N, d.Hi, d.Wi, d.up = dy, self.N, gh, gw, 1
            d.Cin = rup(s.cout, 8) if cin_dy is None else cin_dy
            d.CoutPad = cin_pad_o
            d.bias = None
            d.Cout = s.cin if cout is None else cout
            if s.stride == 1:
                d.w = wbase
                d.KH = d.KW = s.k
                d.stride, d.pad_y, d.pad_x = 1, 1, 1
                d.Gh, d.Gw = gh, gw
                d.Ho, d.Wo, d.oys, d.oyo, d.oxs, d.oxo = gh, gw, 1, 0, 1, 0
            else:
                cls = py * 2 + px
                d.w = wbase + cls * 4 * cin_pad_o * cout_pad_i * esz
                d.KH = d.KW = 2
                d.stride, d.pad_y, d.pad_x = 1, 1 - py, 1 - px
                d.Gh, d.Gw = gh, gw
                d.Ho, d.Wo, d.oys, d.oyo, d.oxs, d.oxo = 2 * gh, 2 * gw, 2, py, 2, px
            d.y, d.y0, d.y1 = y, hip.NULL_VIEW, y1
            d.alpha, d.act = alpha, hip.ACT_NONE
            d.r1, d.r1_nc, d.beta1 = r1, r1_nc, beta1
            d.r2, d.r2_nc, d.beta2 = r2, r2_nc, beta2
            d.accumulate = accumulate
            d.m, d.m_c0, d.m_c1 = m, m_c0, m_c1
            d.m_relu = m_relu
        self.keep.append(arr)
        if len(classes) == 1:
            L.add(hip.lib().ssr_conv2d, C.byref(arr[0]), what=f"conv dgrad {name}")
        else:   # the four parity classes in one launch (csrc/conv.hip, ssr_conv2d_batch)
            L.add(hip.lib().ssr_conv2d_batch, arr, len(classes), what=f"conv dgrad {name}")


def gather_dgrad(cb: "_ConvBuilder", L: Launcher, prefix: str, k: int, x1: View, c1: int, x2: View, c2: int, gh: int,
                 gw: int, y: View, cout: int, **epi):
    """One launch of the gather-form dense-block backward (ParamStore.add_rdb_gather)."""
    st = cb.store
    d = ConvDesc()
    d.dtype = cb.dt
    d.N, d.Hi, d.Wi, d.up = cb.N, gh, gw, 1
    if c1 > 0:
        d.x, d.Cin, d.x2, d.Cin2 = x1, c1, x2, c2
    else:
        d.x, d.Cin, d.x2, d.Cin2 = x2, c2, hip.NULL_VIEW, 0
    d.w = st.gather[(prefix, k)].data_ptr()
    d.CoutPad = st.gather_rows[(prefix, k)]
    d.bias = None
    d.KH = d.KW = 3
    d.stride, d.pad_y, d.pad_x = 1, 1, 1
    d.Gh, d.Gw = gh, gw
    d.Ho, d.Wo, d.oys, d.oyo, d.oxs, d.oxo = gh, gw, 1, 0, 1, 0
    d.Cout = cout
    d.y, d.y0, d.y1 = y, hip.NULL_VIEW, hip.NULL_VIEW
    d.alpha, d.act = 1.0, hip.ACT_NONE
    d.r1, d.r1_nc, d.beta1 = epi.get("r1", hip.NULL_VIEW), epi.get("r1_nc", 0), epi.get("beta1", 0.0)
    d.r2, d.r2_nc, d.beta2 = epi.get("r2", hip.NULL_VIEW), epi.get("r2_nc", 0), epi.get("beta2", 0.0)
    d.accumulate = 0
    d.m, d.m_c0, d.m_c1 = epi.get("m", hip.NULL_VIEW), epi.get("m_c0", 0), epi.get("m_c1", 0)
    cb.keep.append(d)
    L.add(hip.lib().ssr_conv2d, C.byref(d), what=f"conv dgrad-gather {prefix}.slice{k}")
    return d


# split-bf16 mode (fp32x3): LeakyReLU decision fix-up (ssr_conv2d_fixup), OFF by default (SSR_X3_FIXUP=1 switches it on).
# Round-4 experiment, kept as an option with its measurement: recomputing the pre-activations below SSR_X3_FIX_THR exactly
# removes the LOCAL rounding of a layer, but the decisions that differ from an fp32 evaluation's come as much from the 1e-5
# perturbation the layer's INPUTS already carry: 48 -> 38 of 22.8 M generator decisions (9 -> 4 of 5.9 M in D) differ from the
# float64 oracle's, the same with a threshold of 1e-4 and of 1e-3 (gpurun_out/r04m_*; tests/test_gpu_baseline_shapes.py
# "fp32x3-fix"), conv_first.weight 24.5 % -> 11.4 % of elements outside the 1e-3 gate.  Unconditional gradient parity needs
# fp32-accurate products in EVERY layer (the exact fp32 mode; or six bf16 products per fp32 product).
X3_FIXUP = [os.environ.get("SSR_X3_FIXUP", "0") == "1"]
X3_FIX_THR = [float(os.environ.get("SSR_X3_FIX_THR", "1e-4"))]
X3_FIX_CAP = 8192

_DET = [os.environ.get("SSR_DETERMINISTIC", "0") == "1"]


def deterministic() -> bool:
    """Plans built while this is on use fixed-order reductions only (weight gradients of pixel-range splits through per-split
    partial buffers + ssr_wgrad_reduce instead of fp32 atomics into one buffer): two runs give bit-identical parameters.  Switched
    by SSR_DETERMINISTIC=1 or StepConfig.deterministic (train_step.py; option file key `deterministic: true`)."""
    return _DET[0]


class deterministic_mode:
    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _DET[0] = _DET[0], (self.on or _DET[0])

    def __exit__(self, *a):
        _DET[0] = self.prev


_CUS = None


def _device_cus():
    """(compute units, XCDs) of the current device.  The XCD count is not in the device properties: 8 for the 256-CU MI300/MI355
    family (32 CUs per XCD), else one L2 domain is assumed (the XCD-aware item order is a speed hint only)."""
    global _CUS
    if _CUS is None:
        n = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count if torch.cuda.is_available() else 256
        _CUS = (int(n), 8 if n % 32 == 0 and n >= 64 else 1)
    return _CUS


class WgradBatch:
    """Device tables for one batched ssr_conv2d_wgrad launch (layers sharing KHxKW/stride).

    hip.F32X3 (fp32 storage, split-bf16 matrix math; x = hi + lo to 2^-17, the dropped lo*lo term is 2^-16 relative):
    3x3 stride-1 layers: ONE launch of csrc/wgrad_x3.hip over the fp32 buffers - its loader waves split the tiles on the way
    into LDS and the three products (dy_lo.x_hi + dy_hi.x_lo + dy_hi.x_hi) go into one accumulator set (round 5; SSR_X3_WGRAD_FUSED=0
    = the older form); 4x4 stride-2 layers the same since round 6 (wgrad_x3_k4_kernel, 4 x 16-pixel tiles; SSR_X3_WGRAD_FUSED4=0 = the
    older form: the bf16 transpose-read kernel three times over bf16 hi/lo planes of the fp32 buffers, accumulated in the fp32 gradient
    arena, after one ssr_split_bf16_multi pass over the distinct parent buffers)."""

    # by kernel size: the 4x4 layers have few (co, ci) tiles -> more pixel splits (SSR_WGRAD_T3 / _T4: tuning hooks)
    MAX_TILES_PER_ITEM = {3: int(os.environ.get("SSR_WGRAD_T3", "128")), 4: int(os.environ.get("SSR_WGRAD_T4", "64"))}

    def __init__(self, dtype: int, k: int, stride: int, force_atomic: bool = False, det: Optional[bool] = None):
        self.dtype, self.k, self.stride = dtype, k, stride
        self.force_atomic = force_atomic       # another launch accumulates into the same gradients concurrently
        # element type the wgrad kernel reads: bf16 planes of the fp32 buffers in the split passes, the fp32 buffers themselves in the fused 3x3 kernel
        fused = dtype == hip.F32X3 and ((k == 3 and stride == 1 and os.environ.get("SSR_X3_WGRAD_FUSED", "1") == "1")
                                        or (k == 4 and stride == 2 and os.environ.get("SSR_X3_WGRAD_FUSED4", "1") == "1"))
        self.kdt = hip.F32X3 if fused else hip.BF16 if dtype == hip.F32X3 else dtype
        self.layers: List[WgradLayer] = []
        self.items: List[WgradItem] = []
        self.layer_tab = self.item_tab = None
        self.split_tabs: List[torch.Tensor] = []
        self.split_tab = None
        self.splits: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = []
        # deterministic mode: a layer that is split over pixel ranges gets one layer-table entry PER SPLIT whose dw / db point
        # at a partial buffer of that split alone (one writer per gradient element and launch); ssr_wgrad_reduce then adds the
        # partials to the real gradient in split order.  Without it the splits add into one buffer with fp32 atomics (arrival order).
        self.det = deterministic() if det is None else bool(det)      # plans capture the mode when THEY are built (backward plans are built lazily)
        assert not (self.det and force_atomic), "deterministic mode: no second launch may accumulate into the same gradients concurrently"
        self.virtual = set()                  # indices of the per-split entries of self.layers
        self._partials = []                   # (dw_ptr, db_ptr, wsize, bsize, [virtual layer indices])
        self.partial = self.reduce_tab = None
        self.n_reduce = self.max_reduce = 0

    def add(self, x: View, dy: View, N, hi, wi, up, cin, cout, gh, gw, alpha, dw_ptr, cin_w, db_ptr):
        li = len(self.layers)
        self.layers.append(WgradLayer(x, dy, N, hi, wi, up, cin, cout, 1, 1, gh, gw, alpha, dw_ptr, cin_w, db_ptr))
        tiles = hip.lib().ssr_wgrad_tiles(N, gh, gw, self.kdt, self.k)
        # (the fused fp32x3 kernel walks 8 x 16-pixel tiles, half the bf16 kernel's: the same pixels per item)
        splits = max(1, -(-tiles // (self.MAX_TILES_PER_ITEM.get(self.k, 128) * (2 if self.kdt == hip.F32X3 else 1))))
        per = -(-tiles // splits)
        lsp = [li] * splits                   # layer-table entry of split sp
        if self.det and splits > 1:
            lsp = []
            for sp in range(splits):          # dw / db are patched in finalize(), when the partial buffer exists
                lsp.append(len(self.layers))
                self.virtual.add(len(self.layers))
                self.layers.append(WgradLayer(x, dy, N, hi, wi, up, cin, cout, 1, 1, gh, gw, alpha, None, cin_w, None))
            self._partials.append((dw_ptr, db_ptr, cout * cin_w * self.k * self.k, cout if db_ptr else 0, lsp))
        for co0 in range(0, cout, 32):
            for ci0 in range(0, cin_w, hip.lib().ssr_wgrad_ci_tile(self.kdt, self.k)):
                for sp in range(splits):
                    b, e = sp * per, min(tiles, (sp + 1) * per)
                    if b < e:
                        atomic = 1 if ((splits > 1 and not self.det) or self.force_atomic) else 0
                        self.items.append(WgradItem(lsp[sp], co0, ci0, b, e, atomic, 1, lsp[sp], co0))

    def _weight(self, it):
        """MFMA work of an item in (tile, 32-ci half, 32-co plane) units"""
        L = self.layers[it.layer]
        return (it.tile_end - it.tile_begin) * (2 if L.Cin_w - it.ci0 > 32 else 1) * max(1, it.nco)

    # cycles per 16 x 16-pixel tile and per write-out of the bf16 3x3 kernel (tools/wgrad_body_probe.hip, MI355X): the loaders'
    # fetch rate (~12.5 B/clk per CU) bounds both kinds of item, so a single costs 3/4 of a pair for at most half of its products
    COST_TILE = {2: 5900, 1: 4300}
    COST_WRITEOUT = 25000

    @property
    def N_CU(self):
        """compute units of the device the launch will run on (MI355X: 256) - from the device properties, not a constant"""
        return _device_cus()[0]

    @property
    def N_XCD(self):
        return _device_cus()[1]

    def _cost(self, it):
        nco = max(1, it.nco)
        return (it.tile_end - it.tile_begin) * self.COST_TILE[nco] + nco * self.COST_WRITEOUT

    def _makespan(self, items):
        """workgroups are handed to the CUs in index order as CUs become free: list scheduling, longest first"""
        import heapq
        free = [0] * self.N_CU
        for c in sorted((self._cost(it) for it in items), reverse=True):
            heapq.heapreplace(free, free[0] + c)
        return max(free)

    def _balance(self, items):
        """Pixel-range granularity of the items, chosen by simulating the launch: a generator launch is ~2.2 paired items per
        CU, and whole items left 19 % of the CU time idle at the end (the singles, 3/4 of a pair each, are dealt last).  Finer
        items cost a write-out each (fp32 atomics of the whole (co, ci) tile), so the cut is as coarse as the simulation allows."""
        def cut(its, t_pair, t_single):
            out = []
            for it in its:
                n, t = it.tile_end - it.tile_begin, (t_pair if it.nco == 2 else t_single)
                parts = max(1, -(-n // t))
                per = -(-n // parts)
                for b in range(it.tile_begin, it.tile_end, per):
                    out.append(WgradItem(it.layer, it.co0, it.ci0, b, min(it.tile_end, b + per), 1 if parts > 1 else it.atomic,
                                         it.nco, it.layer_b, it.co0_b))
            return out
        best = None
        for t_pair in (128, 64, 32):
            for t_single in (128, 64, 32, 16):
                cand = cut(items, t_pair, t_single)
                m = self._makespan(cand)
                if best is None or m < best[0] * 0.98:         # finer only if it buys 2 %
                    best = (m, cand, t_pair, t_single)
        self.balance_choice = best[2:] + (best[0], self._makespan(items))
        return best[1]

    def _pair(self, items):
        """Two 32-channel blocks of output gradients that are contracted with the SAME input patch share one work item
        (csrc/wgrad_bf16.hip, Wg3: every X fragment read from LDS then feeds two MFMAs).  Same x view, ci0, geometry, pixel
        tiles and the same number of valid input channels; the blocks may belong to different layers — a dense block's
        conv1..conv4 all read x and produce dpre1..dpre4 (rrdbnet_arch.py:37-41) — or be the halves of one 64-output conv."""
        groups = {}
        for it in items:
            L = self.layers[it.layer]
            key = (L.x.p, L.x.cs, L.x.coff, it.ci0, it.tile_begin, it.tile_end, L.N, L.Hi, L.Wi, L.up, L.Gh, L.Gw, L.pad_y, L.pad_x,
                   min(64, L.Cin_w - it.ci0) > 32, min(64, L.Cin - it.ci0), it.atomic)
            groups.setdefault(key, []).append(it)
        out = []
        for g in groups.values():
            for a, b in zip(g[0::2], g[1::2]):
                out.append(WgradItem(a.layer, a.co0, a.ci0, a.tile_begin, a.tile_end, a.atomic, 2, b.layer, b.co0))
            if len(g) % 2:
                out.append(g[-1])
        return out


    def _xcd_order(self, items):
        """Workgroup b runs on XCD b % 8 and every XCD has its own L2.  Items that read the same input buffer over the same
        pixel tiles (a dense block: 14 (conv, co, ci) items over one 192-channel buffer and its gradient) are dealt to ONE
        XCD, next to each other in its queue, so that they walk the tiles together and a line of x / dy is fetched once per
        group instead of once per item (layer-major order spread the 14 over all eight L2s: 3.39 GB per launch for
        ~0.6 GB of distinct lines, profiles/r03k_traffic.json)."""
        weight = self._weight
        groups = {}
        for it in items:
            groups.setdefault((self.layers[it.layer].x.p, it.tile_begin, it.tile_end), []).append(it)
        queues, load = [[] for _ in range(self.N_XCD)], [0] * self.N_XCD
        for g in sorted(groups.values(), key=lambda g: -sum(weight(i) for i in g)):      # heaviest group first, to the least loaded XCD
            q = min(range(self.N_XCD), key=lambda j: (load[j], j))
            queues[q] += sorted(g, key=lambda i: -weight(i))
            load[q] += sum(weight(i) for i in g)
        # block index = 8 * position + xcd has to be dense: level the queue lengths with items from the tails
        while True:
            lo, hi = min(queues, key=len), max(queues, key=len)
            if len(hi) - len(lo) <= 1:
                break
            lo.append(hi.pop())
        queues.sort(key=lambda q: -len(q))
        return [q[j] for j in range(len(queues[0])) for q in queues if j < len(q)]

    def _cost_xcd_order(self, items):
        """Longest items first - what list scheduling over 256 CUs needs - and INSIDE every run of equal-cost items (a generator launch is a few
        cost classes of hundreds of identical items: the dense blocks are all alike) the items that read the same buffers over the same tiles are
        dealt to ONE XCD (block b runs on XCD b % 8), next to each other in its queue: the L2 saving of `_xcd_order` (round 6: 4.43 -> 3.02 GB of
        HBM reads per 3x3 launch) without its cost (+0.7 ms per step: whole groups per XCD unbalance the tail)."""
        items = sorted(items, key=lambda it: -self._cost(it))
        nx = self.N_XCD
        out = []
        i = 0
        while i < len(items):
            c = self._cost(items[i])
            j = i
            while j < len(items) and self._cost(items[j]) == c:
                j += 1
            groups = {}
            for it in items[i:j]:
                groups.setdefault((self.layers[it.layer].x.p, it.tile_begin, it.tile_end), []).append(it)
            queues = [[] for _ in range(nx)]
            for g in sorted(groups.values(), key=len, reverse=True):
                min(queues, key=len).extend(g)
            for pos in range(len(out), len(out) + (j - i)):
                q = queues[pos % nx]
                if not q:
                    q = max(queues, key=len)
                out.append(q.pop(0))
            i = j
        return out

    def _twin(self, v: View, which: int) -> View:
        parent = hip.parent_of(v)
        tw = getattr(parent, "_ssr_bf16_planes", None)     # the planes live and die with the buffer they mirror
        if tw is None:
            tw = (torch.empty_like(parent, dtype=torch.bfloat16), torch.empty_like(parent, dtype=torch.bfloat16))
            parent._ssr_bf16_planes = tw
        if all(parent is not e[0] for e in self.splits):
            self.splits.append((parent, tw[0], tw[1]))
        return View(tw[which].data_ptr(), v.cs, v.coff)

    def _alloc_partials(self):
        """one fp32 buffer [layer][split][dW | db] for the per-split partial gradients + the table of fixed-order sums"""
        if not self._partials:
            return
        dev = torch.device("cuda", torch.cuda.current_device())
        total, plan = 0, []
        for dw_ptr, db_ptr, wsize, bsize, lsp in self._partials:
            stride = rup(wsize + bsize, 4)
            plan.append((total, stride))
            total += stride * len(lsp)
        self.partial = torch.zeros(total, dtype=torch.float32, device=dev)
        base = self.partial.data_ptr()
        red = []
        for (dw_ptr, db_ptr, wsize, bsize, lsp), (off, stride) in zip(self._partials, plan):
            for sp, li in enumerate(lsp):
                self.layers[li].dw = base + 4 * (off + sp * stride)
                self.layers[li].db = (base + 4 * (off + sp * stride + wsize)) if bsize else None
            red.append(ReduceItem(dw_ptr, base + 4 * off, wsize, stride, len(lsp), 0))
            if bsize:
                red.append(ReduceItem(db_ptr, base + 4 * (off + wsize), bsize, stride, len(lsp), 0))
        self.reduce_tab, self.n_reduce = hip.device_table(red), len(red)
        self.max_reduce = max(r.n for r in red)

    def finalize(self):
        if not self.layers:
            return
        self._alloc_partials()
        if hip.lib().ssr_wgrad_co_tile(self.kdt, self.k) == 64 and os.environ.get("SSR_WGRAD_PAIR", "1") == "1":
            self.items = self._pair(self.items)
            if os.environ.get("SSR_WGRAD_BALANCE", "0") == "1":
                self.items = self._balance(self.items)
        # hybrid (default since round 6): longest items first, equal-cost runs dealt to the XCDs by buffer group - 4.44 -> 3.09 GB of HBM reads per
        # 3x3 launch at the same step time (25.36 / 25.42 / 25.40 against 25.40 / 25.46 / 25.35 ms, call r06x3); heavy: longest first only (rounds 3-5);
        # xcd: whole buffer groups per XCD (3.02 GB, +0.7 ms per step: the tail is unbalanced)
        order = os.environ.get("SSR_WGRAD_ORDER", "hybrid")
        self.items = self._xcd_order(self.items) if order == "xcd" else self._cost_xcd_order(self.items) if order == "hybrid" else \
            sorted(self.items, key=lambda it: -self._cost(it))
        self.item_tab = hip.device_table(self.items)
        if self.kdt != hip.BF16 or self.dtype == hip.BF16:
            self.layer_tab = hip.device_table(self.layers)
            return
        # three passes over hi/lo planes; the bias gradient (sum of dy) comes out of the first two (dy_hi + dy_lo)
        for xi, dyi, with_bias in ((0, 0, True), (0, 1, True), (1, 0, False)):
            tab = []
            for L in self.layers:
                tab.append(WgradLayer(self._twin(L.x, xi), self._twin(L.dy, dyi), L.N, L.Hi, L.Wi, L.up, L.Cin, L.Cout, L.pad_y,
                                      L.pad_x, L.Gh, L.Gw, L.alpha, L.dw, L.Cin_w, L.db if with_bias else None))
            self.split_tabs.append(hip.device_table(tab))
        self.layer_tab = self.split_tabs[0]

    def launch(self, L: Launcher):
        if not self.layers:
            return
        lib = hip.lib()
        if self.partial is not None:          # deterministic mode: zeroed partials -> wgrad -> fixed-order sum into the gradient
            L.add(lib.ssr_fill, self.partial.data_ptr(), self.partial.numel(), hip.F32, 0.0, what="zero wgrad partials")
        self._launch_wgrad(L, lib)
        if self.partial is not None:
            L.add(lib.ssr_wgrad_reduce, self.reduce_tab.data_ptr(), self.n_reduce, self.max_reduce, what="wgrad partials -> grad (fixed order)")

    def _launch_wgrad(self, L: Launcher, lib):
        if self.dtype == hip.F32X3 and self.kdt == hip.BF16:
            if all(p.numel() % 8 == 0 for p, _, _ in self.splits):          # every buffer of the batch in ONE launch (ssr_split_bf16_multi)
                if self.split_tab is None:
                    self.split_tab = hip.device_table([hip.SplitItem(p.data_ptr(), h.data_ptr(), l.data_ptr(), p.numel()) for p, h, l in self.splits])
                L.add(lib.ssr_split_bf16_multi, self.split_tab.data_ptr(), len(self.splits), max(p.numel() for p, _, _ in self.splits),
                      what="split bf16")
            else:
                for parent, hi_t, lo_t in self.splits:
                    L.add(lib.ssr_split_bf16, parent.data_ptr(), hi_t.data_ptr(), lo_t.data_ptr(), parent.numel(), what="split bf16")
            for tab in self.split_tabs:
                L.add(lib.ssr_conv2d_wgrad, tab.data_ptr(), self.item_tab.data_ptr(), len(self.items), hip.BF16, self.k, self.k,
                      self.stride, what="wgrad batch (split pass)")
            return
        L.add(lib.ssr_conv2d_wgrad, self.layer_tab.data_ptr(), self.item_tab.data_ptr(), len(self.items),
              self.kdt, self.k, self.k, self.stride, what="wgrad batch")


# =====================================================================================================
# Generator
# =====================================================================================================
def generator_specs(num_in_ch, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32) -> List[ConvSpec]:
    """Layer list in the reference's state_dict order (rrdbnet_arch.py:92-114)."""
    if scale == 2:
        num_in_ch *= 4
    elif scale == 1:
        num_in_ch *= 16
    nf, gc = num_feat, num_grow_ch
    specs = [ConvSpec("conv_first", nf, num_in_ch)]
    for i in range(num_block):
        for j in (1, 2, 3):
            for k in range(1, 5):
                specs.append(ConvSpec(f"body.{i}.rdb{j}.conv{k}", gc, nf + (k - 1) * gc, dgrad_packed=False))
            specs.append(ConvSpec(f"body.{i}.rdb{j}.conv5", nf, nf + 4 * gc, dgrad_packed=False))
    specs.append(ConvSpec("conv_body", nf, nf))
    specs.append(ConvSpec("conv_up1", nf, nf))
    specs.append(ConvSpec("conv_up2", nf, nf))
    if scale in (8, 16):
        specs.append(ConvSpec("conv_up3", nf, nf))
        if scale == 16:
            specs.append(ConvSpec("conv_up4", nf, nf))
    specs.append(ConvSpec("conv_hr", nf, nf))
    specs.append(ConvSpec("conv_last", num_out_ch, nf))
    return specs


class GeneratorPlan:
    """SSR_RRDBNet on NHWC buffers for a fixed (B, H, W)."""

    def _link_rdb_prefetch(self, descs):
        """Fused dense-block launches run back to back: each one can warm L2 with the weights of the next
        (ssr_rdb_desc.w_next).  Measured r01 (B=16): the warm-up loads compete with the weight stream of the running
        launch, dense-block kernels 3.73 -> 4.08 ms per step, so it is opt-in (SSR_RDB_PREFETCH=1)."""
        esz = 2  # bf16
        if os.environ.get("SSR_RDB_PREFETCH", "0") != "1":
            return
        for cur, nxt in zip(descs[:-1], descs[1:]):
            for k in range(5):
                cur.w_next[k] = nxt.w[k]
                cin = self.nf + k * self.gc
                cur.w_next_bytes[k] = 9 * cin * (self.nf if k == 4 else self.gc) * esz

    def __init__(self, store: ParamStore, B: int, H: int, W: int, *, num_in_ch, num_out_ch=3, scale=4, num_feat=64,
                 num_block=23, num_grow_ch=32, training=True, out_buf: Optional[torch.Tensor] = None,
                 d_out_buf: Optional[torch.Tensor] = None, need_input_grad=False, wgrad_atomic: bool = False,
                 bwd_segments: int = 1):
        self.store, self.B, self.dt = store, B, store.dtype
        self.det = deterministic()          # captured now: the weight-gradient batches below follow the mode the plan was built in
        self.scale, self.nf, self.nb, self.gc = scale, num_feat, num_block, num_grow_ch
        self.num_in_ch, self.num_out_ch = num_in_ch, num_out_ch
        self.unshuffle = 2 if scale == 2 else 4 if scale == 1 else 1
        assert H % self.unshuffle == 0 and W % self.unshuffle == 0
        self.Hin, self.Win = H, W
        H, W = H // self.unshuffle, W // self.unshuffle
        self.H, self.W = H, W
        self.cin_eff = num_in_ch * self.unshuffle ** 2
        assert num_feat % 8 == 0 and num_grow_ch % 8 == 0, "num_feat/num_grow_ch must be multiples of 8"
        self.training = training
        tdt, dev = hip.torch_dtype(self.dt), store.device
        nf, gc, nb = self.nf, self.gc, self.nb
        cd = nf + 4 * gc
        z = lambda *s: torch.zeros(*s, dtype=tdt, device=dev)
        self.n_up = {1: 2, 2: 2, 4: 2, 8: 3, 16: 4}[scale] if scale in (1, 2, 4, 8, 16) else 2
        self.up_names = [f"conv_up{i + 1}" for i in range(self.n_up)]
        self.Ho, self.Wo = H * (1 << self.n_up), W * (1 << self.n_up)
        self.xin = z(B, H, W, rup(self.cin_eff, 8))
        n_rdb = 3 * nb
        n_bufs = n_rdb if training else min(n_rdb, 4)
        self.bufs = [z(B, H, W, cd) for _ in range(n_bufs)]
        self.body_out = z(B, H, W, nf)
        self.trunk = z(B, H, W, nf)
        self.ups = [z(B, H << (i + 1), W << (i + 1), nf) for i in range(self.n_up)]
        self.hr = z(B, self.Ho, self.Wo, nf)
        self.out = out_buf if out_buf is not None else z(B, self.Ho, self.Wo, rup(num_out_ch, 8))
        assert self.out.shape[:3] == (B, self.Ho, self.Wo)
        cb = self._cb = _ConvBuilder(store, B)
        self._cb = cb
        # fp32x3, SSR_X3_CHAIN=1 (OFF by default): the dense block's conv1..conv4 (and slices 4..1 of its backward) as one persistent
        # chain launch each (csrc/conv_x3c.hip); the chain's tickets / epoch / flag words live in a small device buffer owned by THIS
        # plan's launches (one stream at a time).  Built and measured in round 6 - and slower than the four
        # launches it replaces: 51.0 against 44 - 48 us per dense block stand-alone (tools/chain_time.py), 26.30 against 25.87 ms per step
        # (call r06i, same box): what the removed launch boundaries save (gap + wave start + first loads) the hand-over spends again
        # (write-through stores drained before the flag, flag visibility, the neighbours' skew) - DESIGN.md lesson 63.
        self._chain_state = None
        if (self.dt == hip.F32X3 and nf == 64 and gc == 32 and not X3_FIXUP[0] and os.environ.get("SSR_X3_CHAIN", "0") == "1"):
            nbytes = int(hip.lib().ssr_conv2d_chain_state_bytes(B, H, W))
            self._chain_state = torch.zeros((nbytes + 3) // 4, dtype=torch.int32, device=dev)
        # ------------------------------------------------------------------ forward
        F = Launcher()
        buf = lambda r: self.bufs[r % n_bufs]
        # feat must survive until conv_body's trunk add; with rotating buffers (inference) keep a copy
        self.feat = self.bufs[0] if n_bufs == n_rdb else z(B, H, W, nf)
        cb.conv(F, "conv_first", view(self.xin), H, W, view(buf(0), 0),
                y0=hip.NULL_VIEW if n_bufs == n_rdb else view(self.feat))
        # bf16, nf = 64, gc = 32: one fused launch per dense block (csrc/rdb_fwd.hip); otherwise 5 convs
        self.fused_rdb = (self.dt == hip.BF16 and nf == 64 and gc == 32 and os.environ.get("SSR_FUSED_RDB", "1") != "0")
        self._rdb_descs = []
        for r in range(n_rdb):
            i, j = divmod(r, 3)
            p = f"body.{i}.rdb{j + 1}"
            cur = buf(r)
            dst = view(self.body_out) if r == n_rdb - 1 else view(buf(r + 1), 0)
            if self.fused_rdb:
                rd = hip.RdbDesc()
                rd.dtype, rd.N, rd.H, rd.W = self.dt, B, H, W
                rd.inp, rd.slices, rd.out, rd.mask = view(cur, 0), view(cur, 0), dst, hip.NULL_VIEW
                for k in range(5):
                    rd.w[k] = store.packed_fwd[f"{p}.conv{k + 1}"].data_ptr()
                    rd.bias[k] = store.ptr(f"{p}.conv{k + 1}.bias")
                rd.w[4] = store.add_repack(f"{p}.conv5", 16).data_ptr()   # conv5 streams in 16-channel half chunks
                if j < 2:
                    rd.alpha5, rd.beta1, rd.r2, rd.beta2 = 0.2, 1.0, hip.NULL_VIEW, 0.0
                else:
                    rd.alpha5, rd.beta1, rd.r2, rd.beta2 = 0.04, 0.2, view(buf(r - 2), 0), 1.0
                self._rdb_descs.append(rd)
                F.add(hip.lib().ssr_rdb_forward, C.byref(rd), what=f"rdb fwd {p}")
                continue
            # conv1..conv4: each reads the channel prefix the earlier ones extend.  fp32x3: ONE persistent launch for the four
            # (csrc/conv_x3c.hip, ssr_conv2d_chain; falls back to four launches by itself where the chain does not qualify)
            tmp = Launcher() if self._chain_state is not None else F
            ds = [cb.conv(tmp, f"{p}.conv{k}", view(cur, 0), H, W, view(cur, nf + (k - 1) * gc), act=hip.ACT_LRELU,
                          cin=nf + (k - 1) * gc) for k in range(1, 5)]
            if tmp is not F:
                self._add_chain(F, ds, f"conv chain fwd {p}.conv1-4")
            if j < 2:   # x5*0.2 + x                                            (rrdbnet_arch.py:44)
                cb.conv(F, f"{p}.conv5", view(cur, 0), H, W, dst, alpha=0.2, r1=view(cur, 0), r1_nc=nf, beta1=1.0,
                        cin=cd)
            else:       # (x5*0.2 + x)*0.2 + x_rrdb                              (rrdbnet_arch.py:44,68)
                cb.conv(F, f"{p}.conv5", view(cur, 0), H, W, dst, alpha=0.04, r1=view(cur, 0), r1_nc=nf, beta1=0.2,
                        r2=view(buf(r - 2), 0), r2_nc=nf, beta2=1.0, cin=cd)
        self._link_rdb_prefetch(self._rdb_descs)
        # feat + conv_body(body)                                                 (rrdbnet_arch.py:124-125)
        cb.conv(F, "conv_body", view(self.body_out), H, W, view(self.trunk), r1=view(self.feat, 0), r1_nc=nf, beta1=1.0)
        src, sh, sw = self.trunk, H, W
        for i, nm in enumerate(self.up_names):  # lrelu(conv(nearest x2))       (rrdbnet_arch.py:127-128)
            cb.conv(F, nm, view(src), sh, sw, view(self.ups[i]), up=2, act=hip.ACT_LRELU)
            src, sh, sw = self.ups[i], sh * 2, sw * 2
        cb.conv(F, "conv_hr", view(src), sh, sw, view(self.hr), act=hip.ACT_LRELU)
        cb.conv(F, "conv_last", view(self.hr), sh, sw, view(self.out))
        self.fwd = F
        if not training:
            return
        assert n_bufs == n_rdb
        # ------------------------------------------------------------------ backward
        # dL/d out (valid: num_out_ch channels); may alias the discriminator's input-gradient buffer
        self.d_out = d_out_buf if d_out_buf is not None else z(B, self.Ho, self.Wo, self.out.shape[-1])
        assert self.d_out.shape == self.out.shape
        self.g_hr = z(B, self.Ho, self.Wo, nf)
        self.g_ups = [z(B, H << (i + 1), W << (i + 1), nf) for i in range(self.n_up)]
        self.g_tmp = [z(B, H << (i + 1), W << (i + 1), nf) for i in range(self.n_up)]
        self.g_trunk = z(B, H, W, nf)
        self.g_body_out = z(B, H, W, nf)
        self.dbufs = [z(B, H, W, cd) for _ in range(n_rdb)]
        self.g_xin = z(B, H, W, self.xin.shape[-1]) if need_input_grad else None
        Bk = Launcher()
        st = store
        # weight gradients: ONE batched launch after the last dgrad (default), or SSR_WGRAD_CHUNKS = n batches, each forked onto
        # a side stream as soon as the gradient buffers it reads are final, beside the remaining (strictly sequential) dgrad chain
        n_chunks = max(1, int(os.environ.get("SSR_WGRAD_CHUNKS", "1")))
        # data parallel (bwd_segments > 1): the backward is cut into SEGMENTS, each = a stretch of the dgrad chain followed IN LINE
        # by the weight-gradient batch of the layers it has finished; the caller exchanges a segment's slice of the gradient arena
        # while the next segment runs (train_step: one all-reduce per segment on the comm stream) — the bucketed overlap of the
        # reference's DDP (README.md:159, options.py:65-81) on flat arenas.  Arena order is the state_dict's (conv_first, body.0 ..
        # body.nb-1, conv_body, conv_up*, conv_hr, conv_last) and the backward walks it from the end, so a segment's parameters
        # are one contiguous range.
        n_seg = max(1, int(bwd_segments))
        if n_seg > 1:
            n_chunks = n_seg
        cuts = {round(n_rdb * q / n_chunks) for q in range(1, n_chunks)}     # close a batch before RDB index r in `cuts`
        batches = [WgradBatch(self.dt, 3, 1, wgrad_atomic, det=self.det)]
        self.bwd_segments = []        # [(Launcher, arena offset, arena elements)]
        seg_state = {"launcher": Bk, "hi": st.numel}

        def add_wg(name, x: View, dy: View, hi, wi, up, gh, gw, alpha=1.0, cin=None):
            s = st.specs[name]
            batches[-1].add(x, dy, B, hi, wi, up, rup(s.cin, 8) if cin is None else cin, s.cout, gh, gw, alpha,
                            st.ptr(name + ".weight", st.grad), s.cin, st.ptr(name + ".bias", st.grad) if s.bias else None)

        def close_batch(next_first_key=None):
            nonlocal Bk
            wgb = batches[-1]
            wgb.finalize()
            if n_seg > 1:
                wgb.launch(Bk)                      # in line: the segment ends when its weight gradients are final
                lo = st.offsets[next_first_key][0] if next_first_key else 0
                hi = seg_state["hi"]
                # the slice [lo, hi) of the gradient arena is what the caller exchanges behind this segment: it must be non-empty
                # and hold EVERY gradient this segment's batch writes (arena order = backward walk from the end; a spec order
                # that stopped matching it would all-reduce the wrong slice silently)
                assert 0 <= lo < hi <= st.numel, (next_first_key, lo, hi)
                g0, esz = st.grad.data_ptr(), st.grad.element_size()
                for iw, Lw in enumerate(wgb.layers):
                    if iw in wgb.virtual:          # per-split partial buffers of the deterministic mode (reduced into the arena in line)
                        continue
                    for ptr in (Lw.dw, Lw.db):
                        if ptr:
                            off = (int(ptr) - g0) // esz
                            assert lo <= off < hi, f"weight gradient at arena offset {off} outside its segment [{lo}, {hi})"
                self.bwd_segments.append((Bk, lo, hi - lo))
                seg_state["hi"] = lo
                Bk = Launcher()
            else:
                sub = Launcher()
                wgb.launch(sub)
                Bk.fork(sub, what="wgrad chunk")
            batches.append(WgradBatch(self.dt, 3, 1, wgrad_atomic, det=self.det))

        Ho, Wo = self.Ho, self.Wo
        last_up = self.ups[-1]
        # conv_last / conv_hr
        add_wg("conv_last", view(self.hr), view(self.d_out), Ho, Wo, 1, Ho, Wo)
        cb.dgrad(Bk, "conv_last", view(self.d_out), Ho, Wo, view(self.g_hr), m=view(self.hr), m_c0=0, m_c1=nf)
        add_wg("conv_hr", view(last_up), view(self.g_hr), Ho, Wo, 1, Ho, Wo)
        cb.dgrad(Bk, "conv_hr", view(self.g_hr), Ho, Wo, view(self.g_ups[-1]), m=view(last_up), m_c0=0, m_c1=nf)
        # upsampling convs, last to first
        for i in reversed(range(self.n_up)):
            nm = self.up_names[i]
            hh, ww = H << (i + 1), W << (i + 1)          # output dims of this conv
            src = self.ups[i - 1] if i > 0 else self.trunk
            add_wg(nm, view(src), view(self.g_ups[i]), hh // 2, ww // 2, 2, hh, ww)
            cb.dgrad(Bk, nm, view(self.g_ups[i]), hh, ww, view(self.g_tmp[i]))
            if i > 0:   # 2x2 sum back to the pre-upsample grid, masked by lrelu'(ups[i-1])
                Bk.add(hip.lib().ssr_nearest2x_bwd, view(self.g_tmp[i]), hip.NULL_VIEW, hip.NULL_VIEW,
                       view(self.g_ups[i - 1]), view(self.ups[i - 1]), self.dt, B, hh // 2, ww // 2, nf,
                       what="nearest2x_bwd")
            else:       # trunk = feat + conv_body(...) is linear: no mask
                Bk.add(hip.lib().ssr_nearest2x_bwd, view(self.g_tmp[i]), hip.NULL_VIEW, hip.NULL_VIEW,
                       view(self.g_trunk), hip.NULL_VIEW, self.dt, B, H, W, nf, what="nearest2x_bwd")
        add_wg("conv_body", view(self.body_out), view(self.g_trunk), H, W, 1, H, W)
        cb.dgrad(Bk, "conv_body", view(self.g_trunk), H, W, view(self.g_body_out))
        # body, last RDB to first
        bwd_descs = []
        for r in reversed(range(n_rdb)):
            i, j = divmod(r, 3)
            p = f"body.{i}.rdb{j + 1}"
            cur, dcur = self.bufs[r], self.dbufs[r]
            d_out_r = view(self.g_body_out) if r == n_rdb - 1 else view(self.dbufs[r + 1], 0)
            if j == 2:
                d_rrdb = d_out_r            # gradient w.r.t. this RRDB's output
                a5, b5 = 0.04, 0.2
            else:
                rr = 3 * i + 2
                d_rrdb = view(self.g_body_out) if rr == n_rdb - 1 else view(self.dbufs[rr + 1], 0)
                a5, b5 = 0.2, 1.0
            fused_bwd = self.fused_rdb and os.environ.get("SSR_FUSED_RDB_BWD", "1") != "0"
            if n_chunks > 1 and (r + 1) in cuts:      # everything recorded so far reads buffers of blocks > r: final by now
                ir, jr = divmod(r + 1, 3)
                close_batch(f"body.{ir}.rdb{jr + 1}.conv1.weight")   # first parameter (arena order) of the blocks finished so far
            store.add_rdb_gather(p, nf, gc, a5, ck0=16 if fused_bwd else 0)
            add_wg(f"{p}.conv5", view(cur, 0), d_out_r, H, W, 1, H, W, alpha=a5, cin=cd)
            for k in (4, 3, 2, 1):
                add_wg(f"{p}.conv{k}", view(cur, 0), view(dcur, nf + (k - 1) * gc), H, W, 1, H, W,
                       cin=nf + (k - 1) * gc)
            if fused_bwd:
                # one launch for the whole dense-block backward (csrc/rdb_fwd.hip, rdb_kernel<true>)
                rd = hip.RdbDesc()
                rd.dtype, rd.N, rd.H, rd.W = self.dt, B, H, W
                rd.inp, rd.slices, rd.out, rd.mask = d_out_r, view(dcur, 0), view(dcur, 0), view(cur, 0)
                for jj in range(5):
                    rd.w[jj] = store.gather[(p, 4 - jj)].data_ptr()
                    rd.bias[jj] = None
                rd.w[4] = store.gather[(p, "k0", 16)].data_ptr()   # slice 0 streams in 16-channel half chunks
                rd.alpha5, rd.beta1 = 1.0, b5
                rd.r2, rd.beta2 = (d_rrdb, 1.0) if j == 0 else (hip.NULL_VIEW, 0.0)
                self._rdb_descs.append(rd)
                bwd_descs.append(rd)
                Bk.add(hip.lib().ssr_rdb_backward, C.byref(rd), what=f"rdb bwd {p}")
                continue
            # gather form: slice k <- one conv over [dpre_{k+1} .. dpre_4 | d_out]; every slice is written once,
            # masked by lrelu'(x_k) in the epilogue, so dgrad needs no read-modify-write
            tmp = Launcher() if self._chain_state is not None else Bk
            ds = [gather_dgrad(cb, tmp, p, k, view(dcur, nf + k * gc), (4 - k) * gc, d_out_r, nf, H, W,
                               view(dcur, nf + (k - 1) * gc), gc, m=view(cur, nf + (k - 1) * gc), m_c0=0, m_c1=gc) for k in (4, 3, 2, 1)]
            if tmp is not Bk:
                self._add_chain(Bk, ds, f"conv chain dgrad-gather {p}.slice4-1")
            kw = dict(r1=d_out_r, r1_nc=nf, beta1=b5)        # the `+ x` path of this RDB         (rrdbnet_arch.py:44)
            if j == 0:                                          # d x_rrdb += d out_rrdb             (rrdbnet_arch.py:68)
                kw.update(r2=d_rrdb, r2_nc=nf, beta2=1.0)
            gather_dgrad(cb, Bk, p, 0, view(dcur, nf), 4 * gc, d_out_r, nf, H, W, view(dcur, 0), nf, **kw)
        self._link_rdb_prefetch(bwd_descs)
        # d feat += d trunk                                                                      (rrdbnet_arch.py:125)
        Bk.add(hip.lib().ssr_add_views, view(self.dbufs[0], 0), view(self.g_trunk), self.dt, B * H * W, nf,
               what="add d_trunk")
        add_wg("conv_first", view(self.xin), view(self.dbufs[0], 0), H, W, 1, H, W, cin=self.xin.shape[-1])
        if need_input_grad:
            cb.dgrad(Bk, "conv_first", view(self.dbufs[0], 0), H, W, view(self.g_xin), cout=self.xin.shape[-1],
                     cin_dy=nf)
        if n_chunks > 1:
            close_batch()
            if n_seg == 1:
                Bk.join()
            batches.pop()
        else:
            batches[0].finalize()
            batches[0].launch(Bk)
        self._wg_batches = batches
        if n_seg > 1:          # the whole backward = the segments in order (eager / instrumented runs; the step runs them one by one)
            whole = Launcher()
            for L, _, _ in self.bwd_segments:
                whole.calls.extend(L.calls)
            self.bwd = whole
            assert sum(n for _, _, n in self.bwd_segments) == st.numel and self.bwd_segments[-1][1] == 0
        else:
            self.bwd = Bk

    def _add_chain(self, L: Launcher, descs, what: str):
        """one ssr_conv2d_chain call for a list of already filled descriptors (copied into one ctypes array)"""
        arr = (ConvDesc * len(descs))()
        for i, d in enumerate(descs):
            C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(ConvDesc))
        self._cb.keep.append(arr)
        L.add(hip.lib().ssr_conv2d_chain, arr, len(descs), self._chain_state.data_ptr(), what=what)

    # ---- boundary: NCHW fp32 tensors of the reference API ----
    def load_input(self, x_nchw: torch.Tensor, scale: float = 1.0):
        assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and x_nchw.is_cuda
        assert tuple(x_nchw.shape) == (self.B, self.num_in_ch, self.Hin, self.Win), x_nchw.shape
        hip.check(hip.lib().ssr_nchw_to_nhwc(x_nchw.data_ptr(), self.B, self.num_in_ch, self.Hin, self.Win,
                                             view(self.xin), self.dt, self.unshuffle, 1, scale, hip.stream_ptr()),
                  "ssr_nchw_to_nhwc")

    def read_output(self, out_nchw: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out_nchw is None:
            out_nchw = torch.empty(self.B, self.num_out_ch, self.Ho, self.Wo, device=self.store.device)
        hip.check(hip.lib().ssr_nhwc_to_nchw(view(self.out), self.dt, out_nchw.data_ptr(), self.B, self.num_out_ch,
                                             self.Ho, self.Wo, hip.stream_ptr()), "ssr_nhwc_to_nchw")
        return out_nchw

    def load_output_grad(self, g_nchw: torch.Tensor):
        hip.check(hip.lib().ssr_nchw_to_nhwc(g_nchw.data_ptr(), self.B, self.num_out_ch, self.Ho, self.Wo,
                                             view(self.d_out), self.dt, 1, 1, 1.0, hip.stream_ptr()),
                  "ssr_nchw_to_nhwc")

    def read_input_grad(self) -> torch.Tensor:
        """dL/dx in NCHW of the *unshuffled* input (scale 4) — inverse unshuffle is done by the caller."""
        g = torch.empty(self.B, self.cin_eff, self.H, self.W, device=self.store.device)
        hip.check(hip.lib().ssr_nhwc_to_nchw(view(self.g_xin), self.dt, g.data_ptr(), self.B, self.cin_eff, self.H,
                                             self.W, hip.stream_ptr()), "ssr_nhwc_to_nchw")
        return g


class SplitGeneratorPlan:
    """Two GeneratorPlans over the two halves of the batch, run as two concurrent launch chains (fork / join, capturable).

    Why: the generator is a chain of ~150 strictly dependent launches, and a fused dense-block launch at B = 32 is 512 workgroups
    on 256 CUs with ONE workgroup per CU (161 KB of LDS): the CUs idle through every launch's ramp and tail — 22 % of the
    dense-block time (SQ_BUSY_CU_CYCLES, profiles/r02b_pmc_sq.json).  Samples are independent, so the two half-batches form two
    independent chains; each launch is then one full round of 256 workgroups and the workgroups of one chain fill the ramps and
    tails of the other.  The halves share the parameter store; their weight-gradient launches accumulate into the same fp32
    gradients concurrently, so every work item uses the atomic write-out (WgradBatch.force_atomic).
    Same arithmetic per sample; only the summation order of the weight gradients over samples differs."""

    def __init__(self, store: ParamStore, B: int, H: int, W: int, *, out_buf: torch.Tensor, d_out_buf: torch.Tensor, parts: int = 2,
                 **kw):
        assert B % parts == 0 and kw.get("training", True)
        h = B // parts
        self.B, self.half, self.store = B, h, store
        self.parts = [GeneratorPlan(store, h, H, W, out_buf=out_buf[i * h:(i + 1) * h], d_out_buf=d_out_buf[i * h:(i + 1) * h],
                                    wgrad_atomic=True, **kw) for i in range(parts)]
        a = self.parts[0]
        self.fused_rdb, self.unshuffle = a.fused_rdb, a.unshuffle
        self.num_in_ch, self.num_out_ch, self.Hin, self.Win, self.Ho, self.Wo = a.num_in_ch, a.num_out_ch, a.Hin, a.Win, a.Ho, a.Wo
        self.out, self.dt = out_buf, a.dt
        self.fwd, self.bwd = Launcher(), Launcher()
        for L, which in ((self.fwd, "fwd"), (self.bwd, "bwd")):
            for p in self.parts[1:]:
                L.fork(getattr(p, which), what="another part of the batch")
            L.calls.extend(getattr(a, which).calls)
            L.join()

    def load_input(self, x_nchw: torch.Tensor, scale: float = 1.0):
        for i, p in enumerate(self.parts):
            p.load_input(x_nchw[i * self.half:(i + 1) * self.half], scale)

    def read_output(self, out_nchw: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out_nchw is None:
            out_nchw = torch.empty(self.B, self.num_out_ch, self.Ho, self.Wo, device=self.store.device)
        for i, p in enumerate(self.parts):
            p.read_output(out_nchw[i * self.half:(i + 1) * self.half])
        return out_nchw


# =====================================================================================================
# Discriminator
# =====================================================================================================
def discriminator_specs(num_in_ch, num_feat=64, in_hw: Optional[Tuple[int, int]] = None, dtype: Optional[int] = None) -> List[ConvSpec]:
    """discriminator_arch.py:28-40.  `in_hw` (input size the store will serve, if known) and `dtype` only steer kernel choice: the
    space-to-depth form of a stride-2 layer works on 32x16-pixel tiles and loses to the pipelined kernel on output grids
    below 24 rows (r01 per-layer times at B=32: conv3, 16x16 grid, 82 vs 66 us; conv1/conv2 67/51 vs 99/87 us).  In the split-bf16
    mode the alternative is the exact fp32 MFMA (conv3: 340 us), so half-empty tiles still win there: 16 rows and up."""
    nf = num_feat
    min_rows = 16 if dtype in (hip.F32X3, hip.F32H3, hip.F32H) else 24
    grid = lambda k: None if in_hw is None or min(in_hw[0] >> k, in_hw[1] >> k) >= min_rows else False
    return _disc_specs(num_in_ch, nf, grid)


def _disc_specs(num_in_ch, nf, grid) -> List[ConvSpec]:
    return [
        ConvSpec("conv0", nf, num_in_ch, 3, 1, True, False),
        ConvSpec("conv1", nf * 2, nf, 4, 2, False, True, s2d=grid(1)),
        ConvSpec("conv2", nf * 4, nf * 2, 4, 2, False, True, s2d=grid(2)),
        ConvSpec("conv3", nf * 8, nf * 4, 4, 2, False, True, s2d=grid(3)),
        ConvSpec("conv4", nf * 4, nf * 8, 3, 1, False, True),
        ConvSpec("conv5", nf * 2, nf * 4, 3, 1, False, True),
        ConvSpec("conv6", nf, nf * 2, 3, 1, False, True),
        ConvSpec("conv7", nf, nf, 3, 1, False, True),
        ConvSpec("conv8", nf, nf, 3, 1, False, True),
        ConvSpec("conv9", 1, nf, 3, 1, True, False),
    ]


class DiscriminatorPlan:
    """SSR_UNetDiscriminatorSN on NHWC buffers for a fixed (B, H, W).

    `forward(x_buf)` plans are built per input buffer (fake / real share all activations, because every
    forward is followed by its own backward before the next forward: ssr_esrgan_model.py:181-227)."""

    def __init__(self, store: ParamStore, B: int, H: int, W: int, *, num_in_ch, num_feat=64, skip_connection=True,
                 training=True):
        assert H % 8 == 0 and W % 8 == 0, "U-Net discriminator needs H, W divisible by 8"
        assert num_feat % 8 == 0
        self.store, self.B, self.H, self.W, self.dt = store, B, H, W, store.dtype
        self.det = deterministic()          # captured now: the backward plans (and their weight-gradient batches) are built lazily
        self.nf, self.skip, self.cd = num_feat, skip_connection, num_in_ch
        self.cdp = rup(num_in_ch, 8)
        tdt, dev = hip.torch_dtype(self.dt), store.device
        nf = num_feat
        z = lambda *s: torch.zeros(*s, dtype=tdt, device=dev)
        H2, W2, H4, W4, H8, W8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        self.x0, self.x1, self.x2, self.x3 = z(B, H, W, nf), z(B, H2, W2, 2 * nf), z(B, H4, W4, 4 * nf), z(B, H8, W8, 8 * nf)
        self.u3, self.a4 = z(B, H4, W4, 8 * nf), z(B, H4, W4, 4 * nf)
        self.u4, self.a5 = z(B, H2, W2, 4 * nf), z(B, H2, W2, 2 * nf)
        self.u5, self.a6 = z(B, H, W, 2 * nf), z(B, H, W, nf)
        self.x6 = z(B, H, W, nf) if skip_connection else self.a6
        self.o7, self.o8 = z(B, H, W, nf), z(B, H, W, nf)
        self.logits = z(B, H, W, 8)
        self._cb = _ConvBuilder(store, B)
        self._fwd_cache: Dict[Tuple[int, int], Launcher] = {}
        self._bwd_cache: Dict[Tuple, Launcher] = {}
        self.training = training
        if training:
            self.d_logits = z(B, H, W, 8)
            self.g_o8, self.g_o7 = z(B, H, W, nf), z(B, H, W, nf)
            self.g_a6, self.g_x6 = z(B, H, W, nf), (z(B, H, W, nf) if skip_connection else None)
            self.g_u5 = z(B, H, W, 2 * nf)
            self.g_a5, self.g_x5 = z(B, H2, W2, 2 * nf), (z(B, H2, W2, 2 * nf) if skip_connection else None)
            self.g_u4 = z(B, H2, W2, 4 * nf)
            self.g_a4, self.g_x4 = z(B, H4, W4, 4 * nf), (z(B, H4, W4, 4 * nf) if skip_connection else None)
            self.g_u3 = z(B, H4, W4, 8 * nf)
            self.g3, self.g2, self.g1, self.g0 = z(B, H8, W8, 8 * nf), z(B, H4, W4, 4 * nf), z(B, H2, W2, 2 * nf), z(B, H, W, nf)
            self.g_in = z(B, H, W, self.cdp)

    def forward_plan(self, x_buf: torch.Tensor) -> Launcher:
        key = x_buf.data_ptr()
        if key in self._fwd_cache:
            return self._fwd_cache[key]
        assert tuple(x_buf.shape) == (self.B, self.H, self.W, self.cdp), (x_buf.shape, self.cdp)
        cb, nf, B, H, W, dt = self._cb, self.nf, self.B, self.H, self.W, self.dt
        H2, W2, H4, W4, H8, W8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        L = Launcher()
        lib = hip.lib()
        LR = hip.ACT_LRELU
        cb.conv(L, "conv0", view(x_buf), H, W, view(self.x0), act=LR, cin=self.cdp)
        cb.conv(L, "conv1", view(self.x0), H, W, view(self.x1), act=LR)
        cb.conv(L, "conv2", view(self.x1), H2, W2, view(self.x2), act=LR)
        cb.conv(L, "conv3", view(self.x2), H4, W4, view(self.x3), act=LR)
        L.add(lib.ssr_bilinear2x_fwd, view(self.x3), hip.NULL_VIEW, view(self.u3), dt, B, H8, W8, 8 * nf,
              what="bilinear u3")
        cb.conv(L, "conv4", view(self.u3), H4, W4, view(self.a4), act=LR)
        # x4 = a4 + x2 is never materialised: the skip add is folded into the bilinear read
        L.add(lib.ssr_bilinear2x_fwd, view(self.a4), view(self.x2) if self.skip else hip.NULL_VIEW, view(self.u4), dt,
              B, H4, W4, 4 * nf, what="bilinear u4")
        cb.conv(L, "conv5", view(self.u4), H2, W2, view(self.a5), act=LR)
        L.add(lib.ssr_bilinear2x_fwd, view(self.a5), view(self.x1) if self.skip else hip.NULL_VIEW, view(self.u5), dt,
              B, H2, W2, 2 * nf, what="bilinear u5")
        if self.skip:   # a6 = lrelu(conv6) kept for the backward mask; x6 = a6 + x0 feeds conv7
            cb.conv(L, "conv6", view(self.u5), H, W, view(self.x6), act=LR, y0=view(self.a6), r1=view(self.x0),
                    r1_nc=nf, beta1=1.0)
        else:
            cb.conv(L, "conv6", view(self.u5), H, W, view(self.a6), act=LR)
        cb.conv(L, "conv7", view(self.x6), H, W, view(self.o7), act=LR)
        cb.conv(L, "conv8", view(self.o7), H, W, view(self.o8), act=LR)
        cb.conv(L, "conv9", view(self.o8), H, W, view(self.logits))
        self._fwd_cache[key] = L
        return L

    def backward_plan(self, x_buf: torch.Tensor, param_grads: bool, input_grad: bool,
                      in_residual: Optional[torch.Tensor] = None) -> Launcher:
        """Backward from self.d_logits.  param_grads=False reproduces the generator phase where D's
        parameters are frozen (ssr_esrgan_model.py:136-137): dgrad only."""
        key = (x_buf.data_ptr(), param_grads, 1 if input_grad else 0, 0 if in_residual is None else in_residual.data_ptr())
        if key in self._bwd_cache:
            return self._bwd_cache[key]
        cb, st, nf, B, H, W, dt = self._cb, self.store, self.nf, self.B, self.H, self.W, self.dt
        H2, W2, H4, W4, H8, W8 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8
        lib = hip.lib()
        L = Launcher()
        wg3, wg4 = WgradBatch(dt, 3, 1, det=self.det), WgradBatch(dt, 4, 2, det=self.det)

        def add_wg(name, x, dy, hi, wi, gh, gw, cin=None):
            if not param_grads:
                return
            s = st.specs[name]
            arena = st.grad_sn if s.sn else st.grad
            (wg3 if s.k == 3 else wg4).add(x, dy, B, hi, wi, 1, rup(s.cin, 8) if cin is None else cin, s.cout, gh, gw,
                                           1.0, st.ptr(st.wkey(name), arena), s.cin,
                                           st.ptr(name + ".bias", st.grad) if s.bias else None)

        NV = hip.NULL_VIEW
        add_wg("conv9", view(self.o8), view(self.d_logits), H, W, H, W)
        cb.dgrad(L, "conv9", view(self.d_logits), H, W, view(self.g_o8), m=view(self.o8), m_c0=0, m_c1=nf)
        add_wg("conv8", view(self.o7), view(self.g_o8), H, W, H, W)
        cb.dgrad(L, "conv8", view(self.g_o8), H, W, view(self.g_o7), m=view(self.o7), m_c0=0, m_c1=nf)
        add_wg("conv7", view(self.x6), view(self.g_o7), H, W, H, W)
        cb.dgrad(L, "conv7", view(self.g_o7), H, W, view(self.g_a6), y1=view(self.g_x6) if self.skip else NV,
                 m=view(self.a6), m_c0=0, m_c1=nf)
        add_wg("conv6", view(self.u5), view(self.g_a6), H, W, H, W)
        cb.dgrad(L, "conv6", view(self.g_a6), H, W, view(self.g_u5))
        L.add(lib.ssr_bilinear2x_bwd, view(self.g_u5), NV, view(self.g_x5) if self.skip else NV, view(self.g_a5),
              view(self.a5), dt, B, H2, W2, 2 * nf, what="bilinear bwd 5")
        add_wg("conv5", view(self.u4), view(self.g_a5), H2, W2, H2, W2)
        cb.dgrad(L, "conv5", view(self.g_a5), H2, W2, view(self.g_u4))
        L.add(lib.ssr_bilinear2x_bwd, view(self.g_u4), NV, view(self.g_x4) if self.skip else NV, view(self.g_a4),
              view(self.a4), dt, B, H4, W4, 4 * nf, what="bilinear bwd 4")
        add_wg("conv4", view(self.u3), view(self.g_a4), H4, W4, H4, W4)
        cb.dgrad(L, "conv4", view(self.g_a4), H4, W4, view(self.g_u3))
        L.add(lib.ssr_bilinear2x_bwd, view(self.g_u3), NV, NV, view(self.g3), view(self.x3), dt, B, H8, W8, 8 * nf,
              what="bilinear bwd 3")
        sk = lambda t: (view(t) if self.skip else NV)
        add_wg("conv3", view(self.x2), view(self.g3), H4, W4, H8, W8)
        cb.dgrad(L, "conv3", view(self.g3), H8, W8, view(self.g2), r1=sk(self.g_x4), r1_nc=4 * nf, beta1=1.0,
                 m=view(self.x2), m_c0=0, m_c1=4 * nf)
        add_wg("conv2", view(self.x1), view(self.g2), H2, W2, H4, W4)
        cb.dgrad(L, "conv2", view(self.g2), H4, W4, view(self.g1), r1=sk(self.g_x5), r1_nc=2 * nf, beta1=1.0,
                 m=view(self.x1), m_c0=0, m_c1=2 * nf)
        add_wg("conv1", view(self.x0), view(self.g1), H, W, H2, W2)
        cb.dgrad(L, "conv1", view(self.g1), H2, W2, view(self.g0), r1=sk(self.g_x6), r1_nc=nf, beta1=1.0,
                 m=view(self.x0), m_c0=0, m_c1=nf)
        add_wg("conv0", view(x_buf), view(self.g0), H, W, H, W, cin=self.cdp)
        if input_grad:
            kw = {}
            if in_residual is not None:
                kw.update(r1=view(in_residual), r1_nc=self.cdp, beta1=1.0)
            cb.dgrad(L, "conv0", view(self.g0), H, W, view(self.g_in), cout=self.cdp, cin_dy=nf, **kw)
        wg3.finalize(); wg4.finalize()
        wg3.launch(L); wg4.launch(L)
        self._keep = getattr(self, "_keep", []) + [wg3, wg4]
        self._bwd_cache[key] = L
        return L
