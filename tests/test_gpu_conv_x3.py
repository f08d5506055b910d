"""GPU parity of the split-bf16 (fp32x3) big-tile kernel, csrc/conv_big_x3.hip: 32x16-pixel x 64-channel workgroup tiles, 16-channel
LDS chunks of [hi | lo] rows, three MFMAs per product, workgroups persistent over images, 16-byte fp32 epilogue.

Every case is checked against an independent float64 torch computation of what the descriptor contract (include/ssr_hip.h,
ssr_conv_desc) prescribes, at 1e-4 of max|ref| (the mode rounds a product at 2^-16: ~1e-5 per layer) - the layers are
nn.Conv2d 3x3 / 4x4 stride 2 and their dgrads of /root/reference/ssr/archs/discriminator_arch.py:28-40,45-69 and
rrdbnet_arch.py:109-112,127-136."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _mods():
    from satlas_super_resolution_amd import engine, hip
    return engine, hip


def _nchw(t, c):
    return t.double().cpu().permute(0, 3, 1, 2)[:, :c]


def _reference(d, w, bias, xb, y_init, r1, r2, m, up=1, cin=None):
    cout = d.Cout
    x = _nchw(xb, w.shape[1] if cin is None else cin)
    if up == 2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    acc = F.conv2d(x, w.double(), None if bias is None else bias.double(), padding=1)
    v = F.leaky_relu(acc, 0.2) if d.act == 1 else acc
    s0 = d.alpha * v
    s1 = s0.clone()
    if d.r1.p:
        s1 += d.beta1 * _nchw(r1, cout)
    if d.r2.p:
        s1 += d.beta2 * _nchw(r2, cout)
    if d.accumulate:
        s1 += _nchw(y_init, cout)
    y = s1
    if d.m.p:
        mm = _nchw(m, cout)
        y = s1 * torch.where(mm > 0, torch.ones_like(mm), torch.full_like(mm, 0.2))
    return {"y": y, "y0": s0, "y1": s1}


VARIANTS = ["plain", "lrelu", "lrelu_r1_y0", "mask", "mask_acc", "mask_r1", "generic"]


def _run_3x3(variant, cin, cout, H, W, B, up=1, impl=4, mode="fp32x3", TOL=TOL):
    engine, hip = _mods()
    dt = hip.dtype_code(mode)
    torch.manual_seed(cin + cout + H + len(variant))
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], dt)
    st.load_state_dict({"c.weight": torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5), "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    cinp = engine.rup(cin, 8)
    Ho, Wo = H * up, W * up
    mk = lambda h, w, c: (torch.randn(B, h, w, c, device="cuda") * 0.5).contiguous()
    xb = mk(H, W, cinp)
    xb[..., cin:] = 0
    r1, r2, m, y_init = mk(Ho, Wo, cout), mk(Ho, Wo, cout), mk(Ho, Wo, cout), mk(Ho, Wo, cout)
    y, y0, y1 = y_init.clone(), torch.zeros_like(y_init), torch.zeros_like(y_init)
    cb = engine._ConvBuilder(st, B)
    L = engine.Launcher()
    kw = dict(act=hip.ACT_LRELU if variant in ("lrelu", "lrelu_r1_y0", "generic") else hip.ACT_NONE, cin=cinp, up=up)
    if variant in ("mask_r1", "lrelu_r1_y0", "generic"):
        kw.update(r1=hip.view(r1), r1_nc=cout, beta1=0.5)
    if variant == "lrelu_r1_y0":
        kw.update(y0=hip.view(y0))
    if variant == "generic":
        kw.update(alpha=0.7, y0=hip.view(y0), r2=hip.view(r2), r2_nc=cout, beta2=-0.25)
    d = cb.conv(L, "c", hip.view(xb), H, W, hip.view(y), **kw)
    if variant.startswith("mask") or variant == "generic":
        d.m, d.m_c0, d.m_c1 = hip.view(m), 0, cout
    if variant in ("mask_acc", "generic"):
        d.accumulate = 1
    if variant == "generic":
        d.y1 = hip.view(y1)
    hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), impl), f"impl {impl}")
    torch.cuda.synchronize()
    ref = _reference(d, st.tensor("c.weight").cpu(), st.tensor("c.bias").cpu(), xb, y_init, r1, r2, m, up, cin=cin)
    assert rel_err(y.cpu().permute(0, 3, 1, 2).double(), ref["y"]) < TOL, (variant, "y", rel_err(y.cpu().permute(0, 3, 1, 2).double(), ref["y"]))
    if variant in ("lrelu_r1_y0", "generic"):
        assert rel_err(y0.cpu().permute(0, 3, 1, 2).double(), ref["y0"]) < TOL, (variant, "y0")
    if variant == "generic":
        assert rel_err(y1.cpu().permute(0, 3, 1, 2).double(), ref["y1"]) < TOL, (variant, "y1")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("cin,cout,H,W,B", [(128, 64, 32, 32, 2), (96, 128, 37, 21, 1), (64, 64, 16, 48, 1), (3, 64, 40, 24, 2), (24, 64, 33, 17, 1)])
def test_big_tile_x3_conv(variant, cin, cout, H, W, B):
    """every epilogue feature, ragged sizes, input widths that do not fill a 16-channel chunk (3 -> 8, 24)"""
    _run_3x3(variant, cin, cout, H, W, B)


def test_big_tile_x3_nearest_x2_read_and_automatic_choice():
    """conv_up1 / conv_up2 read their input through the nearest x2 view (rrdbnet_arch.py:127-128); a grid that fills half the chip
    picks the big tile by itself (ssr_conv2d_variant digit 9), the generator's 32 x 32 body does not"""
    engine, hip = _mods()
    _run_3x3("lrelu", 64, 64, 16, 24, 2, up=2)
    _run_3x3("lrelu", 64, 64, 64, 64, 8, up=1, impl=0)
    st = engine.ParamStore([engine.ConvSpec("a", 64, 64, 3, 1, True, False), engine.ConvSpec("b", 64, 192, 3, 1, True, False)], hip.F32X3)
    cb = engine._ConvBuilder(st, 32)
    L = engine.Launcher()
    big = torch.zeros(32, 128, 128, 64, device="cuda")
    d1 = cb.conv(L, "a", hip.view(big), 128, 128, hip.view(big.clone()))
    body_in, body_out = torch.zeros(32, 32, 32, 192, device="cuda"), torch.zeros(32, 32, 32, 64, device="cuda")
    d2 = cb.conv(L, "b", hip.view(body_in), 32, 32, hip.view(body_out))
    assert hip.lib().ssr_conv2d_variant(C.byref(d1)) % 10 == 9 and hip.lib().ssr_conv2d_variant(C.byref(d2)) % 10 != 9


@pytest.mark.parametrize("groups", ["1", "2"])
def test_big_tile_x3_persistent_over_images(groups, monkeypatch):
    """B = 3 with one / two image groups per tile position: the chunk stream crosses image boundaries (in-stream epilogue with the
    next image's first chunk already in LDS), single-chunk images included"""
    monkeypatch.setenv("SSR_CONV_BIG_G", groups)
    for variant in ("plain", "lrelu_r1_y0", "mask_acc", "generic"):
        _run_3x3(variant, 128, 64, 32, 32, 3)
    _run_3x3("lrelu", 16, 64, 37, 21, 3)          # a single chunk per image
    _run_3x3("lrelu", 8, 64, 20, 20, 3)
    test_stride2_dgrad_x3_parity_classes(128, 64, 32, 32, 3, monkeypatch)
    test_stride2_forward_x3_space_to_depth(64, 128, 32, 32, 3)
    test_stride2_forward_x3_space_to_depth(16, 64, 66, 34, 3)      # one chunk per parity class


@pytest.mark.parametrize("cout,cin,gh,gw,B", [(128, 64, 32, 32, 2), (64, 64, 20, 24, 1), (256, 128, 16, 16, 1)])
def test_stride2_dgrad_x3_parity_classes(cout, cin, gh, gw, B, monkeypatch):
    """dgrad of a 4x4 stride-2 conv (discriminator_arch.py:31-33) = four 2x2 parity-class convs: each class forced onto the big-tile
    kernel, and all four in ONE launch through ssr_conv2d_batch, against torch's conv_transpose2d in float64 with the step's
    epilogue (residual + LeakyReLU-backward mask)"""
    engine, hip = _mods()
    dt = hip.F32X3
    torch.manual_seed(cout + cin + gh)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 4, 2, False, True)], dt)
    w = torch.randn(cout, cin, 4, 4) * (1.0 / (cin * 16) ** 0.5)
    st.load_state_dict({"c.weight_orig": w, "c.weight_u": torch.randn(cout), "c.weight_v": torch.randn(cin * 16)})
    st.spectral_norm(power_iter=False)
    st.pack()
    sigma = float(st.sigma[0])
    mk = lambda h, ww, c: (torch.randn(B, h, ww, c, device="cuda") * 0.5).contiguous()
    dy, r1, m = mk(gh, gw, cout), mk(2 * gh, 2 * gw, cin), mk(2 * gh, 2 * gw, cin)
    ref = F.conv_transpose2d(_nchw(dy, cout), (w / sigma).double(), stride=2, padding=1) + _nchw(r1, cin)
    mm = _nchw(m, cin)
    ref = ref * torch.where(mm > 0, torch.ones_like(mm), torch.full_like(mm, 0.2))
    for mode in ("big_each", "big_batch"):
        y = torch.zeros(B, 2 * gh, 2 * gw, cin, device="cuda")
        cb = engine._ConvBuilder(st, B)
        L = engine.Launcher()
        cb.dgrad(L, "c", hip.view(dy), gh, gw, hip.view(y), r1=hip.view(r1), r1_nc=cin, beta1=1.0, m=hip.view(m), m_c0=0, m_c1=cin)
        fn, args, _ = L.calls[0]
        arr, n = args[0], args[1]
        if mode == "big_batch":
            monkeypatch.setenv("SSR_X3_BIGTILE2", "2")
            hip.check(hip.lib().ssr_conv2d_batch(arr, n, hip.stream_ptr()), "batch")
            monkeypatch.delenv("SSR_X3_BIGTILE2")
        else:
            for k in range(n):
                hip.check(hip.lib().ssr_conv2d_impl(C.byref(arr[k]), hip.stream_ptr(), 4), mode)
        torch.cuda.synchronize()
        e = rel_err(y.cpu().permute(0, 3, 1, 2).double(), ref)
        assert e < TOL, (mode, e)


@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 128, 32, 32, 2), (128, 256, 24, 40, 1), (256, 512, 16, 16, 2), (16, 64, 66, 34, 1)])
def test_stride2_forward_x3_space_to_depth(cin, cout, H, W, B):
    """4x4 stride-2 spectral-norm conv + LeakyReLU (discriminator_arch.py:31-33,45-47) through ssr_conv_desc.s2d in the split mode: a
    2x2 conv over a space-to-depth view gathered by the staging loads, weights packed in that order in 16-channel chunks"""
    engine, hip = _mods()
    dt = hip.F32X3
    torch.manual_seed(cin + cout + H)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 4, 2, False, True)], dt)
    assert st.s2d["c"], "shape should qualify for the space-to-depth path"
    w = torch.randn(cout, cin, 4, 4) * (1.0 / (cin * 16) ** 0.5)
    st.load_state_dict({"c.weight_orig": w, "c.weight_u": torch.randn(cout), "c.weight_v": torch.randn(cin * 16)})
    st.spectral_norm(power_iter=False)
    st.pack()
    sigma = float(st.sigma[0])
    xb = (torch.randn(B, H, W, cin, device="cuda") * 0.5).contiguous()
    yb = torch.zeros(B, H // 2, W // 2, cout, device="cuda")
    cb = engine._ConvBuilder(st, B)
    L = engine.Launcher()
    d = cb.conv(L, "c", hip.view(xb), H, W, hip.view(yb), act=hip.ACT_LRELU)
    assert d.s2d == 1 and hip.lib().ssr_conv2d_variant(C.byref(d)) % 10 == 9
    L.run()
    torch.cuda.synchronize()
    yr = F.leaky_relu(F.conv2d(_nchw(xb, cin), (w / sigma).double(), None, stride=2, padding=1), 0.2)
    e = rel_err(yb.cpu().permute(0, 3, 1, 2).double(), yr)
    assert e < TOL, e


# ---- the producer / MFMA-wave ring kernel of the small grids (csrc/conv_x3q.hip), forced through ssr_conv2d_impl(impl = 6) ----
@pytest.mark.parametrize("variant", ["plain", "lrelu", "mask", "mask_r1", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 32, 32, 32, 2), (160, 32, 32, 32, 1), (192, 64, 32, 32, 2), (96, 32, 21, 37, 1), (24, 64, 9, 7, 1),
                                            (16, 32, 8, 16, 3), (128, 96, 16, 40, 1)])
def test_ring_x3_conv(variant, cin, cout, H, W, B):
    """every epilogue feature, the dense block's widths (4 .. 12 chunks: the ring wraps up to three times), ragged tiles, a single
    chunk (no refill), three 32-channel output groups"""
    _run_3x3(variant, cin, cout, H, W, B, impl=6)


# ---- the register-tiled kernel that took the body over in round 6 (csrc/conv_x3r.hip), forced through ssr_conv2d_impl(impl = 7) ----
@pytest.mark.parametrize("variant", ["plain", "lrelu", "mask", "mask_r1", "lrelu_r1_y0", "mask_acc", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 32, 32, 32, 2), (160, 32, 32, 32, 1), (192, 64, 32, 32, 2), (96, 32, 21, 37, 1), (24, 64, 9, 7, 1),
                                            (16, 32, 8, 16, 3), (8, 32, 8, 16, 1), (128, 96, 16, 40, 1), (320, 64, 16, 16, 1), (40, 20, 24, 24, 1)])
def test_regtile_x3_conv(variant, cin, cout, H, W, B):
    """the three straight-line epilogues of the dense block (bias + LeakyReLU; alpha, bias, one residual; mask) and the generic one
    (y0 / y1 / accumulate / mask + residual), 1 .. 20 chunks (one chunk: a wave without work; twenty: the eight-stage ring wraps twice),
    ragged tiles, a half-filled chunk (24, 40), an output width that is no multiple of 32 (20), three 32-channel groups"""
    _run_3x3(variant, cin, cout, H, W, B, impl=7)


@pytest.mark.parametrize("cin,cout,variant", [(128, 32, "lrelu"), (192, 64, "plain"), (96, 32, "mask")])
def test_regtile_x3_bytes_of_an_image_do_not_depend_on_the_batch(cin, cout, variant):
    """the launch shape picks the tiling (8 x 16-pixel tiles when they fill the chip, 4 x 16 at small launches; 64-channel layers as one
    workgroup per tile or two): every form adds the same products in the same order per output, so an image computed alone (B = 1),
    in a small batch (B = 4: half-height tiles) and inside the benchmarked batch (B = 32: full tiles) has the same bytes"""
    engine, hip = _mods()
    H = W = 32
    torch.manual_seed(cin)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], hip.F32X3)
    st.load_state_dict({"c.weight": torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5), "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    xall = (torch.randn(32, H, W, cin, device="cuda") * 0.5).contiguous()
    mall = (torch.randn(32, H, W, cout, device="cuda") * 0.5).contiguous()
    outs = {}
    for B in (1, 4, 32):
        x, m = xall[:B].contiguous(), mall[:B].contiguous()
        y = torch.zeros(B, H, W, cout, device="cuda")
        cb = engine._ConvBuilder(st, B)
        d = cb.conv(engine.Launcher(), "c", hip.view(x), H, W, hip.view(y), act=hip.ACT_LRELU if variant == "lrelu" else hip.ACT_NONE, cin=cin)
        if variant == "mask":
            d.m, d.m_c0, d.m_c1 = hip.view(m), 0, cout
        assert hip.lib().ssr_conv2d_variant(C.byref(d)) % 10 == 5
        hip.check(hip.lib().ssr_conv2d(C.byref(d), hip.stream_ptr()), "ssr_conv2d")
        torch.cuda.synchronize()
        outs[B] = (y, hip.conv_symbol(d))
    print({B: v[1] for B, v in outs.items()})
    assert len({v[1] for v in outs.values()}) >= 2               # the launch shapes really took different instantiations
    for B in (1, 4):
        assert torch.equal(outs[B][0].view(torch.int32), outs[32][0][:B].view(torch.int32)), (B, outs[B][1], outs[32][1])


@pytest.mark.parametrize("variant", ["plain", "lrelu", "mask", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 32, 32, 32, 2), (192, 64, 32, 32, 2), (160, 32, 32, 32, 32), (96, 32, 21, 37, 1), (24, 64, 9, 7, 1),
                                            (320, 64, 16, 16, 1), (40, 8, 16, 16, 2), (16, 8, 16, 16, 2)])
def test_regtile_exact_fp32_conv(variant, cin, cout, H, W, B):
    """the EXACT fp32 arithmetic mode on the register-tiled kernel (csrc/conv_x3r.hip, template flag EX: fp32 rows in the patch ring,
    v_mfma_f32_32x32x2_f32 - the mode every gate of the reference holds in): the same launch shapes as the split mode, held to the float64
    contract reference at fp32 accuracy (2e-6 of max|ref|: only the summation order differs from an fp32 evaluation)"""
    _run_3x3(variant, cin, cout, H, W, B, impl=7, mode="fp32", TOL=2e-6)


# ---- the fp16-split FORWARD arithmetic (SSR_F32H, mode fp32h, round 6): the split kernels with fp16 pieces (22 bits per operand), weights
#      packed x 2^10, v_mfma_f32_32x32x16_f16.  Held to the float64 contract reference at the EXACT mode's tolerance ----
H_TOL = 2e-6


@pytest.mark.parametrize("variant", ["plain", "lrelu", "mask_r1", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 32, 32, 32, 2), (192, 64, 32, 32, 2), (160, 32, 32, 32, 32), (96, 32, 21, 37, 1), (24, 64, 9, 7, 1),
                                            (320, 64, 16, 16, 1), (40, 20, 24, 24, 1), (128, 32, 32, 32, 4)])
def test_regtile_fp16_split_conv(variant, cin, cout, H, W, B):
    """register-tiled kernel (csrc/conv_x3r.hip, AM = 2): full and half-height tiles, both 64-channel forms, ragged tiles, half-filled chunks"""
    _run_3x3(variant, cin, cout, H, W, B, impl=7, mode="fp32h", TOL=H_TOL)


@pytest.mark.parametrize("variant", ["lrelu", "lrelu_r1_y0", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B,up", [(128, 64, 32, 32, 2, 1), (96, 128, 37, 21, 1, 1), (3, 64, 40, 24, 2, 1), (64, 64, 32, 32, 2, 2)])
def test_big_tile_fp16_split_conv(variant, cin, cout, H, W, B, up):
    """big-tile kernel (csrc/conv_big_x3.hip, conv_bigh3_kernel4): the accumulators start at 2^10 x bias and leave x 2^-10"""
    _run_3x3(variant, cin, cout, H, W, B, up=up, impl=4, mode="fp32h", TOL=H_TOL)


@pytest.mark.parametrize("variant", ["plain", "lrelu", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 32, 16, 16, 2), (72, 96, 12, 20, 1), (16, 64, 8, 8, 3)])
def test_pipelined_fp16_split_conv(variant, cin, cout, H, W, B):
    """the pipelined split kernel (csrc/conv.hip, conv_h3_kernel): where every other shape of the mode ends up"""
    _run_3x3(variant, cin, cout, H, W, B, impl=3, mode="fp32h", TOL=H_TOL)


@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 3, 128, 128, 2), (64, 1, 64, 64, 1), (24, 8, 70, 66, 1)])
def test_thin_output_fp16_rows(cin, cout, H, W, B):
    """thin-output VALU kernel on the mode's packed rows: w = (hi + lo) 2^-10 of the fp16 pieces, fp32 FMAs"""
    _run_3x3("plain", cin, cout, H, W, B, impl=5, mode="fp32h", TOL=H_TOL)


@pytest.mark.parametrize("scale", [2.0 ** -9, 1.0, 2.0 ** 9])
def test_fp16_split_accuracy_over_the_activation_range(scale):
    """what the mode promises and what it does not: activations are staged UNSCALED - an element above 2^-3 keeps 22 bits, a smaller one an
    absolute error of 2^-25.  At ordinary magnitudes (x ~ 0.5) and at x ~ 256 the result is fp32-like; at x ~ 1e-3 the error relative to the
    result grows to ~2^-25 / 1e-3 = 3e-5 - still inside the output gate, but no longer decision-exact (documented in include/ssr_hip.h)"""
    engine, hip = _mods()
    torch.manual_seed(5)
    cin, cout, H, W, B = 128, 32, 32, 32, 2
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], hip.dtype_code("fp32h"))
    w = torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
    st.load_state_dict({"c.weight": w, "c.bias": torch.zeros(cout)})
    st.pack()
    xb = (torch.randn(B, H, W, cin, device="cuda") * 0.5 * scale).contiguous()
    y = torch.zeros(B, H, W, cout, device="cuda")
    d = engine._ConvBuilder(st, B).conv(engine.Launcher(), "c", hip.view(xb), H, W, hip.view(y), cin=cin)
    hip.check(hip.lib().ssr_conv2d(C.byref(d), hip.stream_ptr()), "ssr_conv2d")
    torch.cuda.synchronize()
    ref = F.conv2d(_nchw(xb, cin), w.double(), None, padding=1)
    e = rel_err(y.cpu().permute(0, 3, 1, 2).double(), ref)
    print(f"[fp32h] activation scale {scale:g}: {e:.2e} of max|ref|")
    assert torch.isfinite(y).all()
    assert e < (2e-6 if scale >= 1.0 else 2e-4), e


def test_fp16_split_forward_space_to_depth():
    """4x4 stride-2 layer through the space-to-depth view on the fp16-split big-tile kernel"""
    engine, hip = _mods()
    cin, cout, H, W, B = 64, 128, 32, 32, 2
    torch.manual_seed(11)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 4, 2, False, True)], hip.dtype_code("fp32h"))
    assert st.s2d["c"]
    w = torch.randn(cout, cin, 4, 4) * (1.0 / (cin * 16) ** 0.5)
    st.load_state_dict({"c.weight_orig": w, "c.weight_u": torch.randn(cout), "c.weight_v": torch.randn(cin * 16)})
    st.spectral_norm(power_iter=False)
    st.pack()
    sigma = float(st.sigma[0])
    xb = (torch.randn(B, H, W, cin, device="cuda") * 0.5).contiguous()
    yb = torch.zeros(B, H // 2, W // 2, cout, device="cuda")
    L = engine.Launcher()
    d = engine._ConvBuilder(st, B).conv(L, "c", hip.view(xb), H, W, hip.view(yb), act=hip.ACT_LRELU)
    assert d.s2d == 1 and d.dtype == hip.F32H3 and hip.conv_symbol(d).startswith("conv_bigh3_kernel4<2>")
    L.run()
    torch.cuda.synchronize()
    yr = F.leaky_relu(F.conv2d(_nchw(xb, cin), (w / sigma).double(), None, stride=2, padding=1), 0.2)
    e = rel_err(yb.cpu().permute(0, 3, 1, 2).double(), yr)
    assert e < H_TOL, e


def test_regtile_exact_fp32_is_the_automatic_choice_for_the_body():
    engine, hip = _mods()
    st = engine.ParamStore([engine.ConvSpec("b", 32, 160, 3, 1, True, False)], hip.F32)
    cb = engine._ConvBuilder(st, 32)
    buf = torch.zeros(32, 32, 32, 192, device="cuda")
    d = cb.conv(engine.Launcher(), "b", hip.view(buf, 0), 32, 32, hip.view(buf, 160), act=hip.ACT_LRELU, cin=160)
    assert hip.lib().ssr_conv2d_variant(C.byref(d)) % 10 == 5 and hip.conv_symbol(d) == "conv_x3r_kernel<1, 1, 0, 8, 1>"


def test_regtile_x3_linear_epilogue_with_two_residuals():
    """conv5 of the third dense block of an RRDB: 0.04 (acc + b) + 0.2 x + x_rrdb (rrdbnet_arch.py:44,68) on the straight-line path"""
    engine, hip = _mods()
    B, H, W, cin, cout = 2, 32, 32, 192, 64
    torch.manual_seed(7)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], hip.F32X3)
    w = torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
    st.load_state_dict({"c.weight": w, "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    mk = lambda c: (torch.randn(B, H, W, c, device="cuda") * 0.5).contiguous()
    buf, nxt, rr = mk(192), torch.zeros(B, H, W, 192, device="cuda"), mk(192)
    cb = engine._ConvBuilder(st, B)
    d = cb.conv(engine.Launcher(), "c", hip.view(buf, 0), H, W, hip.view(nxt, 0), alpha=0.04, r1=hip.view(buf, 0), r1_nc=cout, beta1=0.2,
                r2=hip.view(rr, 0), r2_nc=cout, beta2=1.0, cin=cin)
    hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), 7), "impl 7")
    torch.cuda.synchronize()
    ref = 0.04 * F.conv2d(_nchw(buf, cin), w.double(), st.tensor("c.bias").cpu().double(), padding=1) + 0.2 * _nchw(buf, cout) + _nchw(rr, cout)
    e = rel_err(nxt.cpu().permute(0, 3, 1, 2).double()[:, :cout], ref)
    assert e < TOL, e
    assert float(nxt[..., cout:].abs().max()) == 0.0           # the other channels of the destination buffer are untouched


# ---- the dense block's conv1..conv4 / backward slices 4..1 as ONE persistent chain launch (csrc/conv_x3c.hip, ssr_conv2d_chain) ----
def _dense_block_fixture(B, H, W, seed):
    engine, hip = _mods()
    torch.manual_seed(seed)
    nf, gc = 64, 32
    specs = [engine.ConvSpec(f"c{k}", gc, nf + (k - 1) * gc, 3, 1, True, False) for k in range(1, 5)]
    st = engine.ParamStore(specs, hip.F32X3)
    sd = {}
    for k in range(1, 5):
        cin = nf + (k - 1) * gc
        sd[f"c{k}.weight"] = torch.randn(gc, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
        sd[f"c{k}.bias"] = torch.randn(gc) * 0.1
    st.load_state_dict(sd)
    st.pack()
    return engine, hip, st, sd


def _chain_call(hip, descs, state):
    arr = (hip.ConvDesc * len(descs))()
    for i, d in enumerate(descs):
        C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(hip.ConvDesc))
    assert hip.lib().ssr_conv2d_chain_ok(arr, len(descs)) == 1
    hip.check(hip.lib().ssr_conv2d_chain(arr, len(descs), state.data_ptr(), hip.stream_ptr()), "ssr_conv2d_chain")
    return arr


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (3, 21, 37), (32, 32, 32), (1, 8, 16), (2, 64, 64)])
def test_chain_x3_forward_is_bit_identical_to_four_launches(B, H, W):
    """conv1..conv4 of a dense block (rrdbnet_arch.py:37-41: conv k reads the prefix [x | x1 .. x(k-1)] of the block buffer and appends
    x_k) in one persistent launch: same bytes as four launches of the register-tiled kernel, and right against float64 torch.
    Shapes: the body's (also at the benchmarked batch: 256 workgroups, every CU waits on its neighbours), ragged tiles, a single
    tile per image (no neighbour), 32 tiles per image; run TWICE on the same state (tickets / epoch re-arm themselves)"""
    engine, hip, st, sd = _dense_block_fixture(B, H, W, 11 + H)
    nf, gc = 64, 32
    x0 = (torch.randn(B, H, W, nf, device="cuda") * 0.5).contiguous()
    bufs = []
    state = torch.zeros(int(hip.lib().ssr_conv2d_chain_state_bytes(B, H, W) + 3) // 4, dtype=torch.int32, device="cuda")
    for mode in ("four", "chain", "chain_again"):
        buf = torch.zeros(B, H, W, nf + 4 * gc, device="cuda")
        buf[..., :nf] = x0
        cb = engine._ConvBuilder(st, B)
        descs = [cb.conv(engine.Launcher(), f"c{k}", hip.view(buf, 0), H, W, hip.view(buf, nf + (k - 1) * gc), act=hip.ACT_LRELU,
                         cin=nf + (k - 1) * gc) for k in range(1, 5)]
        if mode == "four":
            for d in descs:
                hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), 7), "impl 7")
        else:
            _chain_call(hip, descs, state)
        torch.cuda.synchronize()
        bufs.append(buf)
    assert torch.equal(bufs[0].view(torch.int32), bufs[1].view(torch.int32)), int((bufs[0] != bufs[1]).sum())
    assert torch.equal(bufs[0].view(torch.int32), bufs[2].view(torch.int32))
    cur = _nchw(x0, nf)
    for k in range(1, 5):
        y = F.leaky_relu(F.conv2d(cur, sd[f"c{k}.weight"].double(), sd[f"c{k}.bias"].double(), padding=1), 0.2)
        e = rel_err(_nchw(bufs[1], nf + 4 * gc)[:, nf + (k - 1) * gc:nf + k * gc], y)
        assert e < TOL, (k, e)
        cur = torch.cat([cur, y], 1)


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (32, 32, 32), (1, 24, 40)])
def test_chain_x3_backward_gather_slices(B, H, W):
    """slices 4..1 of the gather-form dense-block backward (engine.gather_dgrad: slice k <- conv over [dpre_(k+1) .. dpre_4 | d_out], masked
    by lrelu'(x_k)): the newest slice comes FIRST in K, the chain walks K last to first.  Against four launches (same products, other
    summation order) and against float64 torch."""
    engine, hip = _mods()
    nf, gc = 64, 32
    torch.manual_seed(5 + H)
    # "later conv" weights W_j [gc or nf][cin_j] of a dense block; only their gathered (rotated, transposed) packing is needed
    names = [f"body.0.rdb1.conv{k}" for k in range(1, 6)]
    specs = [engine.ConvSpec(names[k - 1], gc if k < 5 else nf, nf + (k - 1) * gc, 3, 1, True, False, dgrad_packed=False) for k in range(1, 6)]
    st = engine.ParamStore(specs, hip.F32X3)
    sd = {}
    for k in range(1, 6):
        cin = nf + (k - 1) * gc
        sd[names[k - 1] + ".weight"] = torch.randn(gc if k < 5 else nf, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
        sd[names[k - 1] + ".bias"] = torch.zeros(gc if k < 5 else nf)
    st.load_state_dict(sd)
    st.add_rdb_gather("body.0.rdb1", nf, gc, 0.2)
    st.pack()
    cur = (torch.randn(B, H, W, nf + 4 * gc, device="cuda") * 0.5).contiguous()          # forward activations: the masks
    d_out = (torch.randn(B, H, W, nf, device="cuda") * 0.5).contiguous()
    state = torch.zeros(int(hip.lib().ssr_conv2d_chain_state_bytes(B, H, W) + 3) // 4, dtype=torch.int32, device="cuda")
    outs = []
    for mode in ("four", "chain"):
        dcur = torch.zeros(B, H, W, nf + 4 * gc, device="cuda")
        cb = engine._ConvBuilder(st, B)
        descs = [engine.gather_dgrad(cb, engine.Launcher(), "body.0.rdb1", k, hip.view(dcur, nf + k * gc), (4 - k) * gc, hip.view(d_out), nf, H, W,
                                     hip.view(dcur, nf + (k - 1) * gc), gc, m=hip.view(cur, nf + (k - 1) * gc), m_c0=0, m_c1=gc) for k in (4, 3, 2, 1)]
        if mode == "four":
            for d in descs:
                hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), 7), "impl 7")
        else:
            _chain_call(hip, descs, state)
        torch.cuda.synchronize()
        outs.append(dcur)
    assert rel_err(outs[1].double().cpu(), outs[0].double().cpu()) < 1e-5
    # float64: dpre_k = lrelu'(x_k) * sum_{j > k} conv_transpose(dpre_j, W_j[:, slice k]) with dpre_5 = 0.2 * d_out (conv5's scale folded in)
    g = {5: 0.2 * _nchw(d_out, nf)}
    for k in (4, 3, 2, 1):
        lo, hi = nf + (k - 1) * gc, nf + k * gc
        tot = 0
        for j in range(k + 1, 6):
            wj = sd[names[j - 1] + ".weight"].double()[:, lo:hi]
            tot = tot + F.conv_transpose2d(g[j], wj, padding=1)
        xk = _nchw(cur, nf + 4 * gc)[:, lo:hi]
        g[k] = tot * torch.where(xk > 0, torch.ones_like(xk), torch.full_like(xk, 0.2))
        e = rel_err(_nchw(outs[1], nf + 4 * gc)[:, lo:hi], g[k])
        assert e < TOL, (k, e)


# ---- the thin-output VALU kernel of the fp32 modes (csrc/conv_thin.hip, conv_thin_f32_kernel), forced through ssr_conv2d_impl(impl = 5) ----
@pytest.mark.parametrize("mode", ["fp32x3", "fp32"])
@pytest.mark.parametrize("variant", ["plain", "lrelu", "mask", "mask_acc", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 3, 32, 64, 2), (64, 1, 24, 40, 1), (40, 8, 9, 33, 2), (8, 4, 16, 16, 1), (24, 3, 8, 32, 3), (48, 5, 17, 5, 1)])
def test_thin_output_conv_fp32_modes(mode, variant, cin, cout, H, W, B):
    """conv9 / conv_last / conv0's dgrad (discriminator_arch.py:40,69; rrdbnet_arch.py:113,136): 1 .. 8 output channels, one or two
    32-channel staging passes, a half-filled last 16-channel chunk (24, 40), ragged tiles, every epilogue feature"""
    _run_3x3(variant, cin, cout, H, W, B, impl=5, mode=mode)


def test_thin_output_kernel_is_chosen_by_layer_not_by_batch():
    """the fp32 modes' heads at 128 x 128 run on the VALU kernel at every batch size (an image's bytes must not depend on the batch);
    small grids stay on the MFMA kernels"""
    engine, hip = _mods()
    for B in (1, 32):
        st = engine.ParamStore([engine.ConvSpec("h", 3, 64, 3, 1, True, False)], hip.F32X3)
        cb = engine._ConvBuilder(st, B)
        x, y = torch.zeros(B, 128, 128, 64, device="cuda"), torch.zeros(B, 128, 128, 8, device="cuda")
        d = cb.conv(engine.Launcher(), "h", hip.view(x), 128, 128, hip.view(y), cin=64)
        assert hip.lib().ssr_conv2d_variant(C.byref(d)) % 10 == 7
        x, y = torch.zeros(B, 32, 32, 64, device="cuda"), torch.zeros(B, 32, 32, 8, device="cuda")
        d = cb.conv(engine.Launcher(), "h", hip.view(x), 32, 32, hip.view(y), cin=64)
        assert hip.lib().ssr_conv2d_variant(C.byref(d)) % 10 != 7


@pytest.mark.parametrize("c1,c2,cout", [(96, 64, 32), (32, 64, 32), (128, 64, 64), (16, 64, 32)])
def test_ring_x3_conv_two_input_views(c1, c2, cout):
    """the gather form of the dense-block backward contracts over the channel concatenation [x | x2] of two buffers
    (engine.gather_dgrad): a 16-channel chunk comes from one view, the switch happens between chunks"""
    engine, hip = _mods()
    B, H, W = 2, 32, 32
    torch.manual_seed(c1 + c2 + cout)
    st = engine.ParamStore([engine.ConvSpec("c", cout, c1 + c2, 3, 1, True, False)], hip.F32X3)
    w = torch.randn(cout, c1 + c2, 3, 3) * (1.0 / ((c1 + c2) * 9) ** 0.5)
    st.load_state_dict({"c.weight": w, "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    mk = lambda c: (torch.randn(B, H, W, c, device="cuda") * 0.5).contiguous()
    big1, big2 = mk(192), mk(64)                      # views into wider buffers, as in the step
    x1, x2 = hip.view(big1, 32), hip.view(big2, 0)
    m = mk(cout)
    for impl in (7, 6, 3):
        y = torch.zeros(B, H, W, cout, device="cuda")
        cb = engine._ConvBuilder(st, B)
        L = engine.Launcher()
        d = cb.conv(L, "c", x1, H, W, hip.view(y), cin=c1)
        d.x2, d.Cin2 = x2, c2
        d.m, d.m_c0, d.m_c1 = hip.view(m), 0, cout
        hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), impl), f"impl {impl}")
        torch.cuda.synchronize()
        x = torch.cat([_nchw(big1, 192)[:, 32:32 + c1], _nchw(big2, c2)], 1)
        ref = F.conv2d(x, w.double(), st.tensor("c.bias").cpu().double(), padding=1)
        mm = _nchw(m, cout)
        ref = ref * torch.where(mm > 0, torch.ones_like(mm), torch.full_like(mm, 0.2))
        e = rel_err(y.cpu().permute(0, 3, 1, 2).double(), ref)
        assert e < TOL, (impl, e)


def test_regtile_x3_is_the_automatic_choice_for_the_body(monkeypatch):
    engine, hip = _mods()
    st = engine.ParamStore([engine.ConvSpec("b", 32, 160, 3, 1, True, False)], hip.F32X3)
    cb = engine._ConvBuilder(st, 32)
    L = engine.Launcher()
    buf = torch.zeros(32, 32, 32, 192, device="cuda")
    d = cb.conv(L, "b", hip.view(buf, 0), 32, 32, hip.view(buf, 160), act=hip.ACT_LRELU, cin=160)
    assert hip.lib().ssr_conv2d_variant(C.byref(d)) % 10 == 5            # digit 5 = csrc/conv_x3r.hip (6 = the ring kernel it replaced)


def test_fp16_split_overflow_is_loud_not_wrong():
    """the documented range of the mode (include/ssr_hip.h, SSR_F32H): an activation beyond fp16's 65504 - or a weight beyond 65504 / 2^10 - becomes
    inf in its hi piece and the outputs that touch it come out non-finite; nothing is silently clamped.  (Inside the range: the tests above.)"""
    engine, hip = _mods()
    torch.manual_seed(3)
    cin, cout, H, W, B = 64, 32, 16, 16, 1
    for big_x, big_w in ((True, False), (False, True)):
        st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], hip.dtype_code("fp32h"))
        w = torch.randn(cout, cin, 3, 3) * 0.05
        if big_w:
            w[5, 7, 1, 1] = 100.0                                   # 100 x 2^10 > 65504
        st.load_state_dict({"c.weight": w, "c.bias": torch.zeros(cout)})
        st.pack()
        xb = (torch.randn(B, H, W, cin, device="cuda") * 0.5).contiguous()
        if big_x:
            xb[0, 8, 8, 3] = 1.0e5
        y = torch.zeros(B, H, W, cout, device="cuda")
        d = engine._ConvBuilder(st, B).conv(engine.Launcher(), "c", hip.view(xb), H, W, hip.view(y), cin=cin)
        hip.check(hip.lib().ssr_conv2d(C.byref(d), hip.stream_ptr()), "ssr_conv2d")
        torch.cuda.synchronize()
        bad = ~torch.isfinite(y)
        assert bad.any(), (big_x, big_w)
        if big_x:       # exactly the 3 x 3 neighbourhood of the pixel, every output channel; everything else is untouched by it
            assert bad[0, 7:10, 7:10, :].all() and int(bad.sum()) == 9 * cout
        else:           # the one output channel whose weight overflowed
            assert bad[..., 5].any() and not bad[..., :5].any() and not bad[..., 6:].any()
