"""ctypes binding of libssr_hip.so (the C ABI declared in include/ssr_hip.h).

PyTorch supplies device memory and streams only: every call passes raw ``data_ptr()`` addresses and
the current HIP stream handle.  There is NO CPU / eager fallback: if the library is missing or a
launch fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libssr_hip.so")

F32, BF16, F32X3 = 0, 1, 2    # F32X3: fp32 storage, split-bf16 matrix math (include/ssr_hip.h)
F32H3 = 3                     # SSR_F32H: forward convolutions with fp16-split operands (forward descriptors and forward weight tables only)
ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2
DETERMINISTIC = 0x200          # OR-ed into the dtype of ssr_l1_loss / ssr_bce_logits_loss: per-block loss slots (include/ssr_hip.h)
LOSS_SLOTS = 256
SN_BWD_SLOTS = 64
BILINEAR_FLAT = 0x100         # OR-ed into the dtype of ssr_bilinear2x_fwd / _bwd: per-pixel kernels for this call (include/ssr_hip.h)


class HipLibraryError(RuntimeError):
    pass


class View(C.Structure):
    _fields_ = [("p", C.c_void_p), ("cs", C.c_int32), ("coff", C.c_int32)]


NULL_VIEW = View(None, 0, 0)


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32),
        ("x", View), ("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("up", C.c_int32), ("Cin", C.c_int32),
        ("x2", View), ("Cin2", C.c_int32),
        ("w", C.c_void_p), ("CoutPad", C.c_int32), ("bias", C.c_void_p),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad_y", C.c_int32), ("pad_x", C.c_int32),
        ("Gh", C.c_int32), ("Gw", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("oys", C.c_int32), ("oyo", C.c_int32), ("oxs", C.c_int32),
        ("oxo", C.c_int32), ("Cout", C.c_int32),
        ("y", View), ("y0", View), ("y1", View),
        ("alpha", C.c_float), ("act", C.c_int32),
        ("r1", View), ("r1_nc", C.c_int32), ("beta1", C.c_float),
        ("r2", View), ("r2_nc", C.c_int32), ("beta2", C.c_float),
        ("accumulate", C.c_int32),
        ("m", View), ("m_c0", C.c_int32), ("m_c1", C.c_int32),
        ("s2d", C.c_int32), ("m_relu", C.c_int32),
        ("fix_list", C.c_void_p), ("fix_cap", C.c_int32), ("fix_thr", C.c_float), ("w_ref", C.c_void_p), ("w_ref_sigma", C.c_void_p),
        ("w_ref_cin", C.c_int32),
    ]


class RdbDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("inp", View), ("slices", View), ("out", View), ("mask", View), ("w", C.c_void_p * 5), ("bias", C.c_void_p * 5),
                ("alpha5", C.c_float), ("beta1", C.c_float), ("r2", View), ("beta2", C.c_float),
                ("w_next", C.c_void_p * 5), ("w_next_bytes", C.c_int32 * 5), ("tile", C.c_int32)]


class WgradLayer(C.Structure):
    _fields_ = [
        ("x", View), ("dy", View),
        ("N", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("up", C.c_int32), ("Cin", C.c_int32),
        ("Cout", C.c_int32), ("pad_y", C.c_int32), ("pad_x", C.c_int32), ("Gh", C.c_int32), ("Gw", C.c_int32),
        ("alpha", C.c_float), ("dw", C.c_void_p), ("Cin_w", C.c_int32), ("db", C.c_void_p),
    ]


class WgradItem(C.Structure):
    _fields_ = [("layer", C.c_int32), ("co0", C.c_int32), ("ci0", C.c_int32), ("tile_begin", C.c_int32),
                ("tile_end", C.c_int32), ("atomic", C.c_int32), ("nco", C.c_int32), ("layer_b", C.c_int32), ("co0_b", C.c_int32)]


class PackItem(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("inv_scale", C.c_void_p), ("dst_fwd", C.c_void_p), ("dst_dgrad", C.c_void_p),
        ("Cout", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
        ("CoutPad", C.c_int32), ("CinPad", C.c_int32), ("CinPadO", C.c_int32), ("CoutPadI", C.c_int32),
        ("ck_fwd", C.c_int32), ("ck_dgrad", C.c_int32), ("fwd_s2d", C.c_int32),
    ]


class PackSeg(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("scale", C.c_float), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("ci0", C.c_int32), ("nci", C.c_int32), ("kbase", C.c_int32), ("rows_pad", C.c_int32), ("ck", C.c_int32)]


class SNItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma", C.c_void_p), ("tmp", C.c_void_p),
                ("rows", C.c_int32), ("cols", C.c_int32)]


class SNBwdItem(C.Structure):
    _fields_ = [("dw_sn", C.c_void_p), ("w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma", C.c_void_p),
                ("dw", C.c_void_p), ("tmp", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class ReduceItem(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("n", C.c_int64), ("stride", C.c_int64), ("parts", C.c_int32),
                ("pad_", C.c_int32)]


class SplitItem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p), ("n", C.c_int64)]


class AdamArgs(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("ema", C.c_void_p), ("n", C.c_int64), ("lr", C.c_void_p), ("step", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("ema_decay", C.c_float),
                ("grad_scale", C.c_float)]


# every symbol include/ssr_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ssr_conv2d", "ssr_conv2d_fixup", "ssr_conv2d_batch", "ssr_conv2d_impl", "ssr_conv2d_variant", "ssr_conv2d_symbol", "ssr_conv2d_chain", "ssr_conv2d_chain_ok", "ssr_conv2d_chain_state_bytes", "ssr_conv2d_ck", "ssr_conv2d_s2d_ok", "ssr_rdb_forward", "ssr_rdb_backward", "ssr_rdb_tile_of", "ssr_conv2d_wgrad", "ssr_wgrad_reduce", "ssr_wgrad_tiles", "ssr_wgrad_ci_tile", "ssr_wgrad_co_tile", "ssr_pack_weights", "ssr_pack_dgrad_gather", "ssr_add_views", "ssr_nchw_to_nhwc", "ssr_nhwc_to_nchw",
    "ssr_fill", "ssr_bilinear2x_fwd", "ssr_bilinear2x_bwd", "ssr_nearest2x_bwd", "ssr_spectral_norm",
    "ssr_spectral_norm_bwd", "ssr_usm_sharp", "ssr_l1_loss", "ssr_bce_logits_loss", "ssr_adam_step", "ssr_axpby_f32",
    "ssr_quantize_u8", "ssr_metric_shift_sums", "ssr_metric_ssim_sums", "ssr_split_bf16", "ssr_split_bf16_multi", "ssr_channel_affine", "ssr_relu_maxpool2_fwd", "ssr_relu_maxpool2_bwd",
    "ssr_device_info", "ssr_abi_version",
]

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libssr_hip.so (built by __graft_entry__.build()).  Raises if it is missing: the product
    path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    l.ssr_conv2d.argtypes = [C.POINTER(ConvDesc), vp]
    l.ssr_conv2d_fixup.argtypes = [C.POINTER(ConvDesc), vp]
    l.ssr_conv2d_impl.argtypes = [C.POINTER(ConvDesc), vp, i32]
    l.ssr_conv2d_batch.argtypes = [C.POINTER(ConvDesc), i32, vp]
    l.ssr_conv2d_variant.argtypes = [C.POINTER(ConvDesc)]
    l.ssr_conv2d_symbol.argtypes = [C.POINTER(ConvDesc), C.c_char_p, i32]
    l.ssr_conv2d_chain.argtypes = [C.POINTER(ConvDesc), i32, vp, vp]
    l.ssr_conv2d_chain_ok.argtypes = [C.POINTER(ConvDesc), i32]
    l.ssr_conv2d_chain_state_bytes.argtypes = [i32, i32, i32]
    l.ssr_conv2d_ck.argtypes = [i32, i32]
    l.ssr_conv2d_s2d_ok.argtypes = [i32, i32, i32, i32]
    l.ssr_rdb_forward.argtypes = [C.POINTER(RdbDesc), vp]
    l.ssr_rdb_backward.argtypes = [C.POINTER(RdbDesc), vp]
    l.ssr_rdb_tile_of.argtypes = [C.POINTER(RdbDesc)]
    l.ssr_conv2d_wgrad.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    l.ssr_wgrad_reduce.argtypes = [vp, i32, i64, vp]
    l.ssr_wgrad_ci_tile.argtypes = [i32, i32]
    l.ssr_wgrad_co_tile.argtypes = [i32, i32]
    l.ssr_wgrad_tiles.argtypes = [i32, i32, i32, i32, i32]
    l.ssr_pack_weights.argtypes = [vp, i32, i32, vp]
    l.ssr_pack_dgrad_gather.argtypes = [vp, i32, i32, vp]
    l.ssr_add_views.argtypes = [View, View, i32, i64, i32, vp]
    l.ssr_nchw_to_nhwc.argtypes = [vp, i32, i32, i32, i32, View, i32, i32, i32, f32, vp]
    l.ssr_nhwc_to_nchw.argtypes = [View, i32, vp, i32, i32, i32, i32, vp]
    l.ssr_fill.argtypes = [vp, i64, i32, f32, vp]
    l.ssr_bilinear2x_fwd.argtypes = [View, View, View, i32, i32, i32, i32, i32, vp]
    l.ssr_bilinear2x_bwd.argtypes = [View, View, View, View, View, i32, i32, i32, i32, i32, vp]
    l.ssr_nearest2x_bwd.argtypes = [View, View, View, View, View, i32, i32, i32, i32, i32, vp]
    l.ssr_spectral_norm.argtypes = [vp, i32, i32, i32, i32, vp]
    l.ssr_spectral_norm_bwd.argtypes = [vp, i32, i32, vp]
    l.ssr_l1_loss.argtypes = [View, View, View, i32, i64, i32, f32, vp, vp]
    l.ssr_usm_sharp.argtypes = [vp, vp, i32, i32, i32, f32, f32, f32, vp]
    l.ssr_bce_logits_loss.argtypes = [View, View, i32, i64, f32, f32, vp, vp, vp]
    l.ssr_adam_step.argtypes = [C.POINTER(AdamArgs), vp]
    l.ssr_axpby_f32.argtypes = [f32, vp, f32, vp, i64, vp]
    l.ssr_quantize_u8.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    l.ssr_metric_shift_sums.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    l.ssr_metric_ssim_sums.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    l.ssr_split_bf16.argtypes = [vp, vp, vp, i64, vp]
    l.ssr_split_bf16_multi.argtypes = [vp, i32, i64, vp]
    l.ssr_channel_affine.argtypes = [View, View, i32, i64, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, vp]
    l.ssr_relu_maxpool2_fwd.argtypes = [View, View, i32, i32, i32, i32, i32, vp]
    l.ssr_relu_maxpool2_bwd.argtypes = [View, View, View, i32, i32, i32, i32, i32, i32, vp]
    l.ssr_device_info.argtypes = [C.c_char_p, i32]
    l.ssr_abi_version.argtypes = []
    for s in ABI_SYMBOLS:
        getattr(l, s).restype = i64 if s == "ssr_conv2d_chain_state_bytes" else i32
    _lib = l
    return l


def check(rc: int, what: str):
    if rc != 0:
        raise HipLibraryError(f"{what} failed with code {rc}"
                              + (" (bad descriptor)" if rc == -1 else " (unsupported)" if rc == -2 else " (hipError)"))


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


# "fp32f" (round 6): a MODE of the launch plans, not a dtype of the C ABI - every FORWARD convolution runs in exact fp32 (SSR_F32), every
# backward convolution and weight gradient in the split-bf16 arithmetic (SSR_F32X3); tensors are fp32 in HBM in both, so the two meet in
# the same buffers.  Why: what keeps fp32x3 outside the reference's GRADIENT gate is not the accuracy of its gradients (2e-5 given the
# LeakyReLU decisions) but the decisions themselves - pre-activations rounded at 2^-17 land on the other side of zero ~10x more often
# than in an fp32 evaluation, and each flip moves every upstream gradient.  With the forward exact, the
# decisions are an fp32 evaluation's, and the backward - linear in the incoming gradient given those decisions - keeps its
# matrix-core speed.  Code > 15: never passed to the library (storage_code() gives the code the C ABI sees).
F32F = 16 + F32X3
# "fp32h" (round 6, after fp32f): the same idea at the split-bf16 mode's speed.  What the decisions need is not an exact product but a
# pre-activation as close to the fp32 value as another fp32 summation order would be: FORWARD convolutions on fp16-split operands
# (SSR_F32H: two 11-bit pieces = 22 bits of each operand, three v_mfma_f32_32x32x16_f16 per product, weights pre-scaled by 2^10 so that
# their lo pieces stay in fp16's normal range; include/ssr_hip.h), everything backward in split-bf16 as in fp32x3 / fp32f (gradients of
# 1e-7 do not fit fp16 without a per-tensor scale, and the backward is linear in them given the decisions).  CPU emulation of the
# candidate arithmetics on the full-depth generator (tools/experiments/split_forward_flips.py, 22.8 M decisions): flips against the
# float64 truth - plain fp32 2, bf16 x3 45, fp16 x3 unscaled 23, fp16 x3 with scaled weights 0, bf16 x6 0.
F32H = 32 + F32X3


def storage_code(dt: int) -> int:
    """the dtype code the C ABI sees for tensors / non-conv launches / backward convolutions of mode `dt`"""
    return F32X3 if dt in (F32F, F32H) else dt


def forward_code(dt: int) -> int:
    """the dtype code of the FORWARD convolutions of mode `dt`"""
    return F32 if dt == F32F else F32H3 if dt == F32H else dt


def torch_dtype(dt: int):
    return torch.bfloat16 if dt == BF16 else torch.float32


def dtype_code(dt) -> int:
    if isinstance(dt, str):
        try:
            return {"fp32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16, "fp32x3": F32X3, "bf16x3": F32X3, "fp32f": F32F, "fp32h": F32H}[dt]
        except KeyError:
            raise ValueError(f"unsupported compute dtype {dt!r}") from None
    if dt is torch.float32:
        return F32
    if dt is torch.bfloat16:
        return BF16
    if dt in (F32, BF16, F32X3, F32F, F32H):
        return int(dt)
    raise ValueError(f"unsupported compute dtype {dt!r}")


# NHWC parent tensors by base address: lets a consumer of channel views (the split-bf16 weight-gradient passes) find the
# whole buffer a view points into
import weakref  # noqa: E402

_PARENTS = weakref.WeakValueDictionary()


def parent_of(v: "View") -> torch.Tensor:
    return _PARENTS[v.p]


def view(t: Optional[torch.Tensor], coff: int = 0) -> View:
    """NHWC tensor [..., C] -> channel-sliced view starting at channel `coff`."""
    if t is None:
        return View(None, 0, 0)
    assert t.is_contiguous()
    old = _PARENTS.get(t.data_ptr())
    if old is None or old.numel() < t.numel():      # a leading slice shares its base address with the whole buffer: keep the larger
        _PARENTS[t.data_ptr()] = t
    return View(t.data_ptr(), t.shape[-1], coff)


def device_table(items) -> torch.Tensor:
    """Copy a list of ctypes structs to a device byte tensor (descriptor tables)."""
    arr_t = type(items[0]) * len(items)
    arr = arr_t(*items)
    raw = bytes(memoryview(arr).cast("B"))
    host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    return host.cuda()


def device_info() -> str:
    buf = C.create_string_buffer(256)
    lib().ssr_device_info(buf, 256)
    return buf.value.decode()


def conv_symbol(desc) -> str:
    """the kernel symbol ssr_conv2d launches for `desc`, as rocprofv3 prints it (include/ssr_hip.h, ssr_conv2d_symbol)"""
    buf = C.create_string_buffer(96)
    check(lib().ssr_conv2d_symbol(C.byref(desc), buf, 96), "ssr_conv2d_symbol")
    return buf.value.decode()
