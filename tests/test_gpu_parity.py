"""GPU parity tests (run on the MI355X box: pytest -m gpu).

Every test drives the HIP path through the C ABI (libssr_hip.so via satlas_super_resolution_amd.hip)
and compares with (a) the golden vectors produced by the unmodified reference classes
(tests/golden/*.pt) and (b) the CPU oracle (oracle/esrgan_oracle.py) at larger sizes.

Tolerance (north_star / SURVEY.md D5): |a - ref| <= 1e-3*max|ref| + 1e-3*|ref| in fp32 mode.
"""
from collections import OrderedDict

import pytest
import math

import torch
import torch.nn.functional as F

from conftest import load_golden, parity_close, rel_err

pytestmark = pytest.mark.gpu


def _mods():
    from satlas_super_resolution_amd import engine, hip
    return engine, hip


def _nchw_to_buf(hip, x, buf, dt):
    x = x.contiguous().float().cuda()
    N, C, H, W = x.shape
    hip.check(hip.lib().ssr_nchw_to_nhwc(x.data_ptr(), N, C, H, W, hip.view(buf), dt, 1, 1, 1.0, hip.stream_ptr()),
              "nchw_to_nhwc")


def _buf_to_nchw(hip, buf, C, dt):
    N, H, W, _ = buf.shape
    out = torch.empty(N, C, H, W, device="cuda")
    hip.check(hip.lib().ssr_nhwc_to_nchw(hip.view(buf), dt, out.data_ptr(), N, C, H, W, hip.stream_ptr()),
              "nhwc_to_nchw")
    return out


# ---------------------------------------------------------------------------------------------
# single conv layers vs torch (CPU fp32) — every geometry the path uses
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,k,stride,H,W,B", [
    (64, 32, 3, 1, 32, 32, 2), (192, 64, 3, 1, 32, 32, 1), (3, 64, 3, 1, 20, 12, 2), (64, 3, 3, 1, 16, 48, 1),
    (64, 1, 3, 1, 8, 8, 1), (24, 64, 3, 1, 9, 7, 1), (64, 128, 4, 2, 32, 32, 2), (16, 24, 4, 2, 8, 40, 1),
    (512, 256, 3, 1, 16, 16, 1), (40, 8, 3, 1, 5, 33, 3),
])
@pytest.mark.parametrize("mode", ["fp32", "bf16", "fp32x3"])
def test_conv_layer_fwd_dgrad_wgrad(cin, cout, k, stride, H, W, B, mode):
    engine, hip = _mods()
    dt = hip.dtype_code(mode)
    tdt = hip.torch_dtype(dt)
    torch.manual_seed(cin * 131 + cout)
    spec = engine.ConvSpec("c", cout, cin, k, stride, True, False)
    st = engine.ParamStore([spec], dt)
    w = torch.randn(cout, cin, k, k) * (1.0 / (cin * k * k) ** 0.5)
    b = torch.randn(cout) * 0.1
    st.load_state_dict({"c.weight": w, "c.bias": b})
    st.pack()
    x = torch.randn(B, cin, H, W)
    cinp, coutp = engine.rup(cin, 8), engine.rup(cout, 8)
    Ho, Wo = H // stride, W // stride
    xb = torch.zeros(B, H, W, cinp, dtype=tdt, device="cuda")
    yb = torch.zeros(B, Ho, Wo, coutp, dtype=tdt, device="cuda")
    _nchw_to_buf(hip, x, xb, dt)
    cb = engine._ConvBuilder(st, B)
    L = engine.Launcher()
    cb.conv(L, "c", hip.view(xb), H, W, hip.view(yb), cin=cinp)
    L.run()
    y = _buf_to_nchw(hip, yb, cout, dt).cpu()
    if mode == "bf16":   # reference on bf16-rounded operands, fp32 accumulate
        xr, wr = x.bfloat16().float(), w.bfloat16().float()
        tol = 1e-3      # fp32-accumulated results (weight / bias gradients); stored bf16 outputs are held to one ulp below
    else:
        xr, wr, tol = x, w, (1e-4 if mode == "fp32x3" else 1e-3)      # split operands: 2^-16 per product, ~1e-5 of max|ref| per layer
    xr = xr.clone().requires_grad_(True)
    wr = wr.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=stride, padding=1)
    assert y.shape == yr.shape
    if mode == "bf16":   # identical operands, fp32 accumulation: the stored value is the adjacent bf16 value at worst
        _assert_one_ulp(yb, yr.detach().bfloat16().float(), "fwd")
    else:
        assert rel_err(y, yr) < tol, ("fwd", rel_err(y, yr))
    # dgrad + wgrad
    r = torch.randn_like(yr)
    rr = r.bfloat16().float() if mode == "bf16" else r
    (yr * rr).sum().backward()
    dyb = torch.zeros(B, Ho, Wo, coutp, dtype=tdt, device="cuda")
    dxb = torch.zeros(B, H, W, cinp, dtype=tdt, device="cuda")
    _nchw_to_buf(hip, r, dyb, dt)
    L2 = engine.Launcher()
    cb.dgrad(L2, "c", hip.view(dyb), Ho, Wo, hip.view(dxb), cout=cinp, cin_dy=coutp)
    wg = engine.WgradBatch(dt, k, stride)
    wg.add(hip.view(xb), hip.view(dyb), B, H, W, 1, cinp, cout, Ho, Wo, 1.0, st.ptr("c.weight", st.grad), cin,
           st.ptr("c.bias", st.grad))
    wg.finalize()
    wg.launch(L2)
    st.grad.zero_()
    L2.run()
    dx = _buf_to_nchw(hip, dxb, cin, dt).cpu()
    if mode == "bf16":
        _assert_one_ulp(dxb, xr.grad.bfloat16().float(), "dgrad")
    else:
        assert rel_err(dx, xr.grad) < tol, ("dgrad", rel_err(dx, xr.grad))
    dw = st.tensor("c.weight", st.grad).cpu()
    db = st.tensor("c.bias", st.grad).cpu()
    assert rel_err(dw, wr.grad) < tol, ("wgrad", rel_err(dw, wr.grad))
    assert rel_err(db, br.grad) < tol, ("bgrad", rel_err(db, br.grad))


def test_bilinear_and_nearest_against_torch():
    engine, hip = _mods()
    dt = hip.F32
    torch.manual_seed(5)
    for (B, H, W, Cc) in [(2, 5, 7, 8), (1, 16, 16, 24)]:
        a = torch.randn(B, Cc, H, W, requires_grad=True)
        b2 = torch.randn(B, Cc, H, W)
        ab, bb = torch.zeros(B, H, W, Cc, device="cuda"), torch.zeros(B, H, W, Cc, device="cuda")
        yb = torch.zeros(B, 2 * H, 2 * W, Cc, device="cuda")
        _nchw_to_buf(hip, a.detach(), ab, dt)
        _nchw_to_buf(hip, b2, bb, dt)
        hip.check(hip.lib().ssr_bilinear2x_fwd(hip.view(ab), hip.view(bb), hip.view(yb), dt, B, H, W, Cc,
                                               hip.stream_ptr()), "bil")
        yr = F.interpolate(a + b2, scale_factor=2, mode="bilinear", align_corners=False)
        assert rel_err(_buf_to_nchw(hip, yb, Cc, dt), yr) < 1e-5
        r = torch.randn_like(yr)
        (yr * r).sum().backward()
        rb = torch.zeros(B, 2 * H, 2 * W, Cc, device="cuda")
        gb = torch.zeros(B, H, W, Cc, device="cuda")
        _nchw_to_buf(hip, r, rb, dt)
        hip.check(hip.lib().ssr_bilinear2x_bwd(hip.view(rb), hip.NULL_VIEW, hip.NULL_VIEW, hip.view(gb),
                                               hip.NULL_VIEW, dt, B, H, W, Cc, hip.stream_ptr()), "bilb")
        assert rel_err(_buf_to_nchw(hip, gb, Cc, dt), a.grad) < 1e-5
        # nearest x2 backward = 2x2 sum
        a2 = torch.randn(B, Cc, H, W, requires_grad=True)
        yn = F.interpolate(a2, scale_factor=2, mode="nearest")
        (yn * r).sum().backward()
        hip.check(hip.lib().ssr_nearest2x_bwd(hip.view(rb), hip.NULL_VIEW, hip.NULL_VIEW, hip.view(gb),
                                              hip.NULL_VIEW, dt, B, H, W, Cc, hip.stream_ptr()), "nb")
        assert rel_err(_buf_to_nchw(hip, gb, Cc, dt), a2.grad) < 1e-6


@pytest.mark.parametrize("B,H,W,Cc,fp32", [(2, 16, 16, 128, False), (1, 7, 21, 64, False), (3, 9, 4, 192, False), (2, 32, 32, 256, False),
                                             (2, 16, 16, 128, True), (1, 7, 21, 32, True), (2, 9, 4, 96, True)])
def test_bilinear_tile_kernels_match_the_per_pixel_kernels(B, H, W, Cc, fp32):
    """layers whose channels are whole groups of eight 16-byte vectors (the U-Net discriminator, discriminator_arch.py:47,52,57)
    run the LDS-tile kernels: same arithmetic in the same order as the per-pixel kernels -> identical bytes in bf16, last-bit
    differences at most in fp32 storage (fused multiply-add choices of the compiler); and against torch."""
    engine, hip = _mods()
    lib = hip.lib()
    g = torch.Generator().manual_seed(B * 100 + H)
    dev = torch.device("cuda:0")
    tdt, DT, ity, tol = (torch.float32, hip.F32, torch.int32, 2e-6) if fp32 else (torch.bfloat16, hip.BF16, torch.int16, 4e-3)
    rnd = lambda *s: torch.randn(*s, generator=g).to(tdt).to(dev)
    a, b2 = rnd(B, H, W, Cc), rnd(B, H, W, Cc)
    dy, r, m = rnd(B, 2 * H, 2 * W, Cc), rnd(B, H, W, Cc), rnd(B, H, W, Cc)

    def run(flat, with_b, with_rm):
        dtf = DT | (hip.BILINEAR_FLAT if flat else 0)        # per-call flag (include/ssr_hip.h), no library state
        y = torch.zeros(B, 2 * H, 2 * W, Cc, dtype=tdt, device=dev)
        hip.check(lib.ssr_bilinear2x_fwd(hip.view(a), hip.view(b2) if with_b else hip.NULL_VIEW, hip.view(y), dtf, B, H, W, Cc,
                                         hip.stream_ptr()), "bil fwd")
        gx, g1 = torch.zeros_like(a), torch.zeros_like(a)
        hip.check(lib.ssr_bilinear2x_bwd(hip.view(dy), hip.view(r) if with_rm else hip.NULL_VIEW, hip.view(g1) if with_rm else hip.NULL_VIEW,
                                         hip.view(gx), hip.view(m) if with_rm else hip.NULL_VIEW, dtf, B, H, W, Cc,
                                         hip.stream_ptr()), "bil bwd")
        torch.cuda.synchronize()
        return y, gx, g1

    for with_b, with_rm in ((True, True), (False, False)):
        y0, gx0, g10 = run(True, with_b, with_rm)
        y1, gx1, g11 = run(False, with_b, with_rm)
        for u, v, what in ((y0, y1, "forward"), (gx0, gx1, "backward"), (g10, g11, "backward, value before the mask")):
            if fp32:      # the two kernels may contract different multiply-add pairs: last-bit differences in fp32 storage
                assert float((u - v).abs().max()) <= 4e-6 * float(u.abs().max()), what
            else:         # bf16 storage: the final rounding absorbs them - identical bytes
                assert torch.equal(u.view(ity), v.view(ity)), what
        # and against torch (fp32 on the bf16 inputs; one bf16 rounding at the end)
        xin = (a.float() + (b2.float() if with_b else 0)).permute(0, 3, 1, 2).cpu().requires_grad_(True)
        yr = F.interpolate(xin, scale_factor=2, mode="bilinear", align_corners=False)
        assert rel_err(y1.float().permute(0, 3, 1, 2).cpu(), yr.detach()) < tol
        yr.backward(dy.float().permute(0, 3, 1, 2).cpu())
        gref = xin.grad + (r.float().permute(0, 3, 1, 2).cpu() if with_rm else 0)
        if with_rm:
            assert rel_err(g11.float().permute(0, 3, 1, 2).cpu(), gref) < tol
            mm = m.float().permute(0, 3, 1, 2).cpu()
            gref = gref * torch.where(mm > 0, torch.ones_like(mm), torch.full_like(mm, 0.2))
        assert rel_err(gx1.float().permute(0, 3, 1, 2).cpu(), gref) < tol
        assert float(gx1.float().abs().max()) > 0.1


def test_index_maps_bit_exact():
    """pixel_unshuffle and nearest-upsample index arithmetic must be bit exact (north_star)."""
    engine, hip = _mods()
    fx = load_golden("index_maps")
    dt = hip.F32
    for key, ref in fx.items():
        if key.startswith("unshuffle"):
            c, hh, hw, s = map(int, key.split("_")[1:])
            x = torch.arange(c * hh * hw, dtype=torch.float32).view(1, c, hh, hw).cuda()
            c2 = c * s * s
            buf = torch.zeros(1, hh // s, hw // s, engine.rup(c2, 8), device="cuda")
            hip.check(hip.lib().ssr_nchw_to_nhwc(x.data_ptr(), 1, c, hh, hw, hip.view(buf), dt, s, 1, 1.0,
                                                 hip.stream_ptr()), "unshuffle")
            got = buf[..., :c2].permute(0, 3, 1, 2).to(torch.int64).cpu()
            assert torch.equal(got, ref), key
        elif key.startswith("nearest"):
            f = int(key[len("nearest")])
            x = torch.arange(2 * 3 * 5, dtype=torch.float32).view(1, 2, 3, 5).cuda()
            buf = torch.zeros(1, 3 * f, 5 * f, 8, device="cuda")
            hip.check(hip.lib().ssr_nchw_to_nhwc(x.data_ptr(), 1, 2, 3, 5, hip.view(buf), dt, 1, f, 1.0,
                                                 hip.stream_ptr()), "nearest")
            got = buf[..., :2].permute(0, 3, 1, 2).to(torch.int64).cpu()
            assert torch.equal(got, ref), key


# ---------------------------------------------------------------------------------------------
# whole networks vs golden vectors from the reference classes
# ---------------------------------------------------------------------------------------------
def _run_generator(fx, mode="fp32", training=True):
    engine, hip = _mods()
    dt = hip.dtype_code(mode)
    kw = fx["kwargs"]
    st = engine.ParamStore(engine.generator_specs(**kw), dt)
    st.load_state_dict(fx["state_dict"])
    B, _, H, W = fx["x"].shape
    plan = engine.GeneratorPlan(st, B, H, W, training=training, need_input_grad=training, **kw)
    st.pack()
    plan.load_input(fx["x"].cuda().contiguous())
    plan.fwd.run()
    return engine, hip, st, plan, plan.read_output()


@pytest.mark.parametrize("name", ["g_tiny_ragged", "g_mid_24ch", "g_scale2", "g_scale1"])
def test_generator_golden_forward_backward(name):
    fx = load_golden(name)
    engine, hip, st, plan, y = _run_generator(fx)
    assert parity_close(y, fx["y"]), rel_err(y, fx["y"])
    plan.load_output_grad(fx["r"].cuda().contiguous())
    st.grad.zero_()
    plan.bwd.run()
    worst = 0.0
    for k, g in fx["grads"].items():
        got = st.tensor(k, st.grad)
        assert parity_close(got, g), (k, rel_err(got, g))
        worst = max(worst, rel_err(got, g))
    if fx["kwargs"]["scale"] == 4:
        dx = plan.read_input_grad()
        assert parity_close(dx, fx["dx"]), rel_err(dx, fx["dx"])


def test_generator_golden_inference_plan_matches():
    """training=False uses 4 rotating dense-block buffers; result must equal the training plan."""
    fx = load_golden("g_tiny_ragged")
    _, _, _, _, y = _run_generator(fx, training=False)
    assert parity_close(y, fx["y"]), rel_err(y, fx["y"])


def test_generator_bf16_mode_close():
    fx = load_golden("g_mid_24ch")
    _, _, _, _, y = _run_generator(fx, mode="bf16", training=False)
    assert rel_err(y, fx["y"]) < 3e-2, rel_err(y, fx["y"])


@pytest.mark.parametrize("name", ["d_tiny", "d_in6_noskip"])
def test_discriminator_golden_forward_backward(name):
    engine, hip = _mods()
    fx = load_golden(name)
    dt = hip.F32
    kw = fx["kwargs"]
    st = engine.ParamStore(engine.discriminator_specs(kw["num_in_ch"], kw["num_feat"]), dt)
    st.load_state_dict(fx["state_dict_before"])
    B, C, H, W = fx["x"].shape
    plan = engine.DiscriminatorPlan(st, B, H, W, **kw)
    xb = torch.zeros(B, H, W, plan.cdp, device="cuda")
    _nchw_to_buf(hip, fx["x"], xb, dt)
    st.spectral_norm(power_iter=True)
    st.pack()
    plan.forward_plan(xb).run()
    y = _buf_to_nchw(hip, plan.logits, 1, dt)
    assert parity_close(y, fx["y"]), rel_err(y, fx["y"])
    for n in st.sn_names:
        assert rel_err(st.u[n], fx["state_dict_after"][n + ".weight_u"]) < 1e-4, n
        assert rel_err(st.v[n], fx["state_dict_after"][n + ".weight_v"]) < 1e-4, n
    _nchw_to_buf(hip, fx["r"], plan.d_logits, dt)
    st.grad.zero_()
    st.grad_sn.zero_()
    plan.backward_plan(xb, param_grads=True, input_grad=True).run()
    st.spectral_norm_backward()
    for k, g in fx["grads"].items():
        got = st.tensor(k, st.grad)
        assert parity_close(got, g), (k, rel_err(got, g))
    dx = _buf_to_nchw(hip, plan.g_in, C, dt)
    assert parity_close(dx, fx["dx"]), rel_err(dx, fx["dx"])
    # eval mode: sigma from stored u, v without power iteration
    st.spectral_norm(power_iter=False)
    st.pack()
    plan.forward_plan(xb).run()
    y_eval = _buf_to_nchw(hip, plan.logits, 1, dt)
    assert parity_close(y_eval, fx["y_eval"]), rel_err(y_eval, fx["y_eval"])


# ---------------------------------------------------------------------------------------------
# the fused G+D train step vs the reference-driven golden step and vs the CPU oracle
# ---------------------------------------------------------------------------------------------
def _update_close(got, ref, p0, lr_steps, what, upd_tol=2e-2):
    """Post-step parameters are compared on the *update*.  Adam normalises the gradient, so an element
    whose gradient is at rounding-noise level gets a +-lr step whose sign is noise in ANY fp32
    implementation; allow a small fraction (<0.1 %) of such elements, bounded by the maximal Adam step."""
    upd_ref, upd = ref - p0, got - p0
    err = (upd - upd_ref).abs()
    tight = upd_tol * upd_ref.abs().max() + 3e-7 * ref.abs().max() + 1e-9
    bad = (err > tight).float().mean().item()
    assert bad <= 1e-3, (what, "fraction of elements off", bad)
    assert err.max() <= 2.1 * lr_steps + tight, (what, float(err.max()))


def _check_step_against(fx_logs, fx_g0, fx_d0, g_final, d_final, g_ema_final, ts, n_iters=2, lr=1e-4):
    for k, v in g_final.items():
        _update_close(ts.g_store.tensor(k).cpu(), v, fx_g0[k], lr * n_iters, ("G", k))
    sd_d = ts.d_store.state_dict()
    for k, v in d_final.items():
        got = sd_d[k].cpu()
        if k.endswith("_u") or k.endswith("_v"):
            assert rel_err(got, v) < 1e-3, ("D buffer", k, rel_err(got, v))
        else:
            _update_close(got, v, fx_d0[k], lr * n_iters, ("D", k))
    ema = ts.ema_state_dict()
    for k, v in g_ema_final.items():   # ema - g0 = (1-decay) * sum of parameter updates
        _update_close(ema[k].cpu(), v, fx_g0[k], lr * n_iters * 1e-3 * n_iters, ("EMA", k))


@pytest.mark.parametrize("name", ["step_tiny", "step_tiny_feedlr"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_train_step_golden(name, use_graph):
    from satlas_super_resolution_amd.train_step import ESRGANTrainStep, StepConfig
    fx = load_golden(name)
    cfg = StepConfig(l1_weight=fx["l1_weight"], gan_weight=fx["gan_weight"], lr_g=fx["lr"], lr_d=fx["lr"],
                     betas=tuple(fx["betas"]), ema_decay=fx["ema_decay"], feed_disc_lr=fx["feed_disc_lr"])
    B = fx["data"][0][0].shape[0]
    ts = ESRGANTrainStep(fx["g_kwargs"], fx["d_kwargs"], B, 8, 8, "fp32", cfg, use_graph=use_graph)
    ts.load_state(fx["g0"], fx["d0"])
    for it, (lr, gt) in enumerate(fx["data"], start=1):
        ts.feed_data(lr.cuda(), gt.cuda())
        ts.step(it)
        log = ts.log()
        for k, v in fx["logs"][it - 1].items():
            assert abs(log[k] - v) <= 1e-3 * max(1.0, abs(v)), (it, k, log[k], v)
        if it == 1:
            # gradients of iteration 1 are still in the arenas only for D (G's were consumed by Adam but
            # not cleared): compare both
            for k, g in fx["g_grads_iter1"].items():
                got = ts.g_store.tensor(k, ts.g_store.grad)
                assert parity_close(got, g), ("g grad", k, rel_err(got, g))
            for k, g in fx["d_grads_iter1"].items():
                got = ts.d_store.tensor(k, ts.d_store.grad)
                assert parity_close(got, g), ("d grad", k, rel_err(got, g))
    out = ts.output()
    assert parity_close(out, fx["output_last"]), rel_err(out, fx["output_last"])
    _check_step_against(fx["logs"], fx["g0"], fx["d0"], fx["g_final"], fx["d_final"], fx["g_ema_final"], ts)


def test_train_step_vs_oracle_baseline_shape():
    """BASELINE.json configs[1] shape at reduced depth/batch (1xS2 RGB, 32x32 -> 128x128; nf=64, gc=32, nb=2,
    B=2) against the CPU oracle, two iterations."""
    from oracle import esrgan_oracle as O
    from satlas_super_resolution_amd.train_step import ESRGANTrainStep, StepConfig
    g_kw = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=2, num_grow_ch=32)
    d_kw = dict(num_in_ch=3, num_feat=64, skip_connection=True)
    g0 = O.generator_init(seed=1, **g_kw)
    d0 = O.discriminator_init(3, 64, seed=2)
    torch.manual_seed(3)
    data = [(torch.rand(2, 3, 32, 32), torch.rand(2, 3, 128, 128)) for _ in range(2)]
    orc = O.ESRGANOracle(g0, d0, O.StepConfig())
    ts = ESRGANTrainStep(g_kw, d_kw, 2, 32, 32, "fp32", StepConfig(), use_graph=True)
    ts.load_state(g0, d0)
    for it, (lr, gt) in enumerate(data, start=1):
        ref_log = orc.step(lr, gt, it)
        ts.feed_data(lr.cuda(), gt.cuda())
        ts.step(it)
        log = ts.log()
        for k, v in ref_log.items():
            assert abs(log[k] - v) <= 1e-3 * max(1.0, abs(v)), (it, k, log[k], v)
        if it == 1:
            for k, g in orc.g_grads.items():
                got = ts.g_store.tensor(k, ts.g_store.grad)
                assert parity_close(got, g), ("g grad", k, rel_err(got, g))
            for k, g in orc.d_grads.items():
                got = ts.d_store.tensor(k, ts.d_store.grad)
                assert parity_close(got, g), ("d grad", k, rel_err(got, g))
    assert parity_close(ts.output(), orc.output)
    _check_step_against(None, g0, d0, orc.g, {k: orc.d[k] for k in orc.d}, orc.g_ema, ts)


# ---------------------------------------------------------------------------------------------
# the drop-in boundary: nn.Module plugins under torch.autograd, and the model plugin
# ---------------------------------------------------------------------------------------------
def test_arch_plugins_autograd_matches_golden():
    from satlas_super_resolution_amd.archs.rrdbnet_arch import SSR_RRDBNet
    from satlas_super_resolution_amd.archs.discriminator_arch import SSR_UNetDiscriminatorSN
    fx = load_golden("g_tiny_ragged")
    net = SSR_RRDBNet(**fx["kwargs"])
    net.load_state_dict(fx["state_dict"], strict=True)
    net = net.cuda().train()
    x = fx["x"].cuda().requires_grad_(True)
    y = net(x)
    assert parity_close(y, fx["y"])
    (y * fx["r"].cuda()).sum().backward()
    assert parity_close(x.grad, fx["dx"])
    for k, p in net.named_parameters():
        assert parity_close(p.grad, fx["grads"][k]), k
    with torch.no_grad():
        assert parity_close(net(fx["x"].cuda()), fx["y"])       # inference plan
    # discriminator: train-mode forward updates u/v in place; frozen params -> dgrad only
    fd = load_golden("d_tiny")
    d = SSR_UNetDiscriminatorSN(**fd["kwargs"])
    d.load_state_dict(fd["state_dict_before"], strict=True)
    d = d.cuda().train()
    xd = fd["x"].cuda().requires_grad_(True)
    yd = d(xd)
    assert parity_close(yd, fd["y"])
    (yd * fd["r"].cuda()).sum().backward()
    assert parity_close(xd.grad, fd["dx"])
    for k, p in d.named_parameters():
        assert parity_close(p.grad, fd["grads"][k]), k
    sd = d.state_dict()
    for n in ("conv1", "conv8"):
        assert rel_err(sd[n + ".weight_u"], fd["state_dict_after"][n + ".weight_u"]) < 1e-4
    d.eval()
    with torch.no_grad():
        assert parity_close(d(fd["x"].cuda()), fd["y_eval"])


def test_model_plugin_runs_reference_style_loop(tmp_path):
    """SSRESRGANModel driven the way /root/reference/ssr/train.py:106-126 drives it, YAML-shaped opt."""
    from satlas_super_resolution_amd import models  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    fx = load_golden("step_tiny")
    opt = {
        "model_type": "SSRESRGANModel", "scale": 4, "manual_seed": 0, "is_train": True, "dist": False,
        "l1_gt_usm": False, "percep_gt_usm": False, "gan_gt_usm": False, "feed_disc_lr": False,
        "network_g": dict(type="SSR_RRDBNet", **fx["g_kwargs"]),
        "network_d": dict(type="SSR_UNetDiscriminatorSN", **fx["d_kwargs"]),
        "path": {"models": str(tmp_path / "models"), "training_states": str(tmp_path / "states")},
        "train": {"ema_decay": 0.999, "optim_g": {"type": "Adam", "lr": 1e-4, "weight_decay": 0, "betas": [0.9, 0.99]},
                  "optim_d": {"type": "Adam", "lr": 1e-4, "weight_decay": 0, "betas": [0.9, 0.99]},
                  "scheduler": {"type": "MultiStepLR", "milestones": [400000], "gamma": 0.5},
                  "pixel_opt": {"type": "L1Loss", "loss_weight": 1.0, "reduction": "mean"},
                  "gan_opt": {"type": "GANLoss", "gan_type": "vanilla", "real_label_val": 1.0,
                              "fake_label_val": 0.0, "loss_weight": 0.1},
                  "net_d_iters": 1, "net_d_init_iters": 0},
    }
    model = build_model(opt)
    for it, (lr, gt) in enumerate(fx["data"], start=1):
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_data({"lr": (lr * 255).round().to(torch.uint8), "hr": (gt * 255).round().to(torch.uint8)})
        if it == 1:
            model.ts.load_state(fx["g0"], fx["d0"])
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert set(log) == {"l_g_pix", "l_g_gan", "l_d_real", "out_d_real", "l_d_fake", "out_d_fake"}
        for k, v in fx["logs"][it - 1].items():      # inputs were quantised to uint8: loose
            assert abs(log[k] - v) < 2e-2, (k, log[k], v)
    assert model.get_current_learning_rate() == [1e-4]
    model.test()
    vis = model.get_current_visuals()
    assert vis["result"].shape == (2, 3, 32, 32) and vis["lr"].shape == (2, 6, 8, 8)
    model.save(0, 2)
    ck = torch.load(tmp_path / "models" / "net_g_2.pth")
    assert set(ck) == {"params", "params_ema"} and "conv_first.weight" in ck["params_ema"]
    assert set(torch.load(tmp_path / "models" / "net_d_2.pth")) == {"params"}
    with pytest.raises(KeyError):            # the reference indexes opt['l1_gt_usm'] directly
        bad = dict(opt); bad.pop("l1_gt_usm")
        m2 = build_model(bad)
        m2.feed_data({"lr": torch.zeros(2, 6, 8, 8, dtype=torch.uint8), "hr": torch.zeros(2, 3, 32, 32, dtype=torch.uint8)})
        m2.optimize_parameters(1)


def test_train_step_bf16_mode_tracks_fp32_oracle():
    """bf16 is a throughput mode (bf16 storage/MFMA inputs, fp32 accumulate, fp32 master weights):
    it cannot meet the 1e-3 gate; hold it to 5e-2 on losses/gradients of the BASELINE-shaped step instead."""
    from oracle import esrgan_oracle as O
    from satlas_super_resolution_amd.train_step import ESRGANTrainStep, StepConfig
    g_kw = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=2, num_grow_ch=32)
    d_kw = dict(num_in_ch=3, num_feat=64, skip_connection=True)
    g0 = O.generator_init(seed=1, **g_kw)
    d0 = O.discriminator_init(3, 64, seed=2)
    torch.manual_seed(3)
    lr, gt = torch.rand(2, 3, 32, 32), torch.rand(2, 3, 128, 128)
    orc = O.ESRGANOracle(g0, d0, O.StepConfig())
    ref_log = orc.step(lr, gt, 1)
    ts = ESRGANTrainStep(g_kw, d_kw, 2, 32, 32, "bf16", StepConfig(), use_graph=False)
    ts.load_state(g0, d0)
    ts.feed_data(lr.cuda(), gt.cuda())
    ts.step(1)
    log = ts.log()
    for k, v in ref_log.items():
        assert abs(log[k] - v) <= 5e-2 * max(1.0, abs(v)), (k, log[k], v)
    worst = 0.0
    for k, g in orc.g_grads.items():
        worst = max(worst, rel_err(ts.g_store.tensor(k, ts.g_store.grad), g))
    for k, g in orc.d_grads.items():
        worst = max(worst, rel_err(ts.d_store.tensor(k, ts.d_store.grad), g))
    assert worst < 8e-2, worst
    assert rel_err(ts.output(), orc.output) < 3e-2


def test_fused_rdb_forward_matches_per_conv_path(monkeypatch):
    """csrc/rdb_fwd.hip (one launch per dense block, bf16) against the per-conv bf16 path and the fp32 oracle,
    on a ragged image size (tile edges) and on the BASELINE tile size."""
    from oracle import esrgan_oracle as O
    engine, hip = _mods()
    kw = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=2, num_grow_ch=32)
    sd = O.generator_init(seed=4, **kw)
    for name in list(sd):      # non-zero biases
        if name.endswith(".bias"):
            sd[name] = torch.randn_like(sd[name]) * 0.05
    for (B, H, W) in [(2, 32, 32), (1, 19, 27)]:
        torch.manual_seed(H)
        x = torch.rand(B, 3, H, W)
        outs, bufs = {}, {}
        for fused in ("1", "0"):
            monkeypatch.setenv("SSR_FUSED_RDB", fused)
            st = engine.ParamStore(engine.generator_specs(**kw), hip.BF16)
            st.load_state_dict(sd)
            plan = engine.GeneratorPlan(st, B, H, W, training=True, **kw)
            assert plan.fused_rdb == (fused == "1")
            st.pack()
            plan.load_input(x.cuda())
            plan.fwd.run()
            outs[fused] = plan.read_output().cpu()
            bufs[fused] = [b.float().cpu() for b in plan.bufs]
        with torch.no_grad():
            ref = O.generator_forward(sd, x)
        for r, (a, b) in enumerate(zip(bufs["1"], bufs["0"])):    # every dense buffer (x, x1..x4 of every RDB)
            assert rel_err(a, b) < 2e-2, ("dense buffer", r, rel_err(a, b))
        assert rel_err(outs["1"], outs["0"]) < 2e-2
        assert rel_err(outs["1"], ref) < 3e-2, rel_err(outs["1"], ref)


def test_fused_rdb_backward_matches_per_conv_path(monkeypatch):
    """rdb_kernel<true> (one launch per dense-block backward, bf16) against the per-conv gather path (bf16) behind
    the SAME (fused) forward, so that both see identical LeakyReLU masks: every gradient buffer and every
    parameter gradient; ragged size included."""
    from oracle import esrgan_oracle as O
    engine, hip = _mods()
    kw = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=2, num_grow_ch=32)
    sd = O.generator_init(seed=4, **kw)
    monkeypatch.setenv("SSR_FUSED_RDB", "1")
    for (B, H, W) in [(2, 32, 32), (1, 19, 27)]:
        torch.manual_seed(H)
        x = torch.rand(B, 3, H, W)
        gout = torch.randn(B, 3, 4 * H, 4 * W)
        res = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("SSR_FUSED_RDB_BWD", fused)
            st = engine.ParamStore(engine.generator_specs(**kw), hip.BF16)
            st.load_state_dict(sd)
            plan = engine.GeneratorPlan(st, B, H, W, training=True, **kw)
            st.pack()
            plan.load_input(x.cuda())
            plan.fwd.run()
            plan.load_output_grad(gout.cuda())
            st.grad.zero_()
            plan.bwd.run()
            res[fused] = ([b.float().cpu() for b in plan.dbufs], st.grad.clone().cpu())
        for r, (a, b) in enumerate(zip(res["1"][0], res["0"][0])):
            for c0, c1 in ((0, 64), (64, 96), (96, 128), (128, 160), (160, 192)):
                e = rel_err(a[..., c0:c1], b[..., c0:c1])
                assert e < 3e-2, ("gradient buffer", r, "channels", c0, c1, e, (B, H, W))
        assert rel_err(res["1"][1], res["0"][1]) < 3e-2



def _desc_reference(d, w, bias, xb, y_init, r1=None, r2=None, m=None, up=1):
    """What the fused epilogue contract (include/ssr_hip.h, ssr_conv_desc) prescribes for descriptor `d`, computed on the CPU in
    fp32 from the bf16 operand values, each output rounded to bf16 once:  s0 = alpha*act(conv + bias) -> y0;
    s1 = s0 + beta1*r1 + beta2*r2 (+ y_old) -> y1;  y = s1 * lrelu'(m).  The kernel families are checked against THIS (an
    independent torch computation), not against each other."""
    nchw = lambda t, c: t.float().cpu().permute(0, 3, 1, 2)[:, :c]
    cout = d.Cout
    x = nchw(xb, w.shape[1])
    if up == 2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    acc = F.conv2d(x, w.bfloat16().float(), bias, padding=1)
    v = F.leaky_relu(acc, 0.2) if d.act == 1 else acc
    s0 = d.alpha * v
    s1 = s0.clone()
    if d.r1.p:
        s1[:, :d.r1_nc] += d.beta1 * nchw(r1, cout)[:, :d.r1_nc]
    if d.r2.p:
        s1[:, :d.r2_nc] += d.beta2 * nchw(r2, cout)[:, :d.r2_nc]
    if d.accumulate:
        s1 = s1 + nchw(y_init, cout)
    y = s1.clone()
    if d.m.p:
        mm = nchw(m, cout)
        y[:, d.m_c0:d.m_c1] = (s1 * torch.where(mm > 0, torch.ones_like(mm), torch.full_like(mm, 0.2)))[:, d.m_c0:d.m_c1]
    rb = lambda t: t.bfloat16().float()
    return {"y": rb(y), "y0": rb(s0), "y1": rb(s1)}


def _assert_one_ulp(got_nhwc, ref_nchw, what):
    """bf16 outputs: every element within one bf16 ulp of the reference, at most 0.5 % differ at all (small tensors), no
    systematic error."""
    from oracle import layerwise as LW
    rep = LW.Report()
    rep.add(str(what), got_nhwc.float().cpu().permute(0, 3, 1, 2)[:, :ref_nchw.shape[1]], ref_nchw)
    n = ref_nchw.numel()          # tiny tensors: allow three single-ulp flips outright
    rep.check_bf16(max_ulps=1.0, max_frac=max(5e-3, 3.0 / n), tol_mean=max(2e-5, 3 * 2.0 ** -7 / n))


@pytest.mark.parametrize("cin,cout,H,W,B,up", [(64, 64, 16, 32, 2, 1), (8, 64, 9, 21, 1, 1), (32, 8, 16, 16, 1, 1),
                                              (64, 128, 8, 16, 1, 1), (24, 32, 5, 7, 2, 2)])
def test_weight_stationary_conv_vs_float64_contract_reference(cin, cout, H, W, B, up):
    """csrc/conv_ws.hip forced through ssr_conv2d_impl(impl=1) AND the pipelined kernel (impl=3) on the same descriptor, each held
    to an independent float64 torch evaluation of the descriptor contract (_desc_reference) within one bf16 ulp - not to each other -
    with the whole epilogue contract switched on: bias, LeakyReLU, alpha, both residuals, dual
    outputs, in-place accumulate, LReLU-backward mask, nearest x2 read."""
    import ctypes as C
    engine, hip = _mods()
    dt, tdt = hip.BF16, torch.bfloat16
    torch.manual_seed(cin + cout + H)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], dt)
    st.load_state_dict({"c.weight": torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5), "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    Ho, Wo = H * up, W * up
    mk = lambda h, w, c: (torch.randn(B, h, w, c, device="cuda") * 0.5).to(tdt).contiguous()
    xb, r1, r2, m = mk(H, W, cin), mk(Ho, Wo, cout), mk(Ho, Wo, cout), mk(Ho, Wo, cout)
    y_init = mk(Ho, Wo, cout)
    outs = {}
    for impl in (1, 3):
        y, y0, y1 = y_init.clone(), torch.zeros_like(y_init), torch.zeros_like(y_init)
        cb = engine._ConvBuilder(st, B)
        L = engine.Launcher()
        d = cb.conv(L, "c", hip.view(xb), H, W, hip.view(y), up=up, act=hip.ACT_LRELU, alpha=0.7, y0=hip.view(y0),
                    r1=hip.view(r1), r1_nc=cout, beta1=0.5, r2=hip.view(r2), r2_nc=cout, beta2=-0.25, cin=cin)
        d.y1 = hip.view(y1)
        d.accumulate = 1
        d.m, d.m_c0, d.m_c1 = hip.view(m), 0, cout
        hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), impl), f"impl {impl}")
        torch.cuda.synchronize()
        outs[impl] = (y.float().cpu(), y0.float().cpu(), y1.float().cpu())
        ref = _desc_reference(d, st.tensor("c.weight").cpu(), st.tensor("c.bias").cpu(), xb, y_init, r1, r2, m, up)
        for t, nm in ((y, "y"), (y0, "y0"), (y1, "y1")):
            _assert_one_ulp(t, ref[nm], (impl, nm))


@pytest.mark.parametrize("variant", ["plain", "lrelu", "mask", "mask_acc", "mask_r1", "mask_y1"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 64, 16, 32, 2), (16, 64, 9, 21, 1), (64, 128, 8, 16, 1)])
def test_weight_stationary_conv_lean_epilogues(variant, cin, cout, H, W, B):
    """The branch-free epilogue instantiations of csrc/conv_ws.hip (the combinations the train step launches:
    forward LeakyReLU, plain, and the dgrad forms mask / mask+accumulate / mask+residual) against the pipelined
    kernel on the same descriptor."""
    import ctypes as C
    engine, hip = _mods()
    dt, tdt = hip.BF16, torch.bfloat16
    torch.manual_seed(cin + cout + H + len(variant))
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], dt)
    st.load_state_dict({"c.weight": torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5), "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    mk = lambda c: (torch.randn(B, H, W, c, device="cuda") * 0.5).to(tdt).contiguous()
    xb, r1, m, y_init = mk(cin), mk(cout), mk(cout), mk(cout)
    outs = {}
    for impl in (1, 3):
        y, y1 = y_init.clone(), torch.zeros_like(y_init)
        cb = engine._ConvBuilder(st, B)
        L = engine.Launcher()
        kw = dict(act=hip.ACT_LRELU if variant == "lrelu" else hip.ACT_NONE, cin=cin)
        if variant == "mask_r1":
            kw.update(r1=hip.view(r1), r1_nc=cout, beta1=0.5)
        d = cb.conv(L, "c", hip.view(xb), H, W, hip.view(y), **kw)
        if variant.startswith("mask"):
            d.m, d.m_c0, d.m_c1 = hip.view(m), 0, cout
        if variant == "mask_acc":
            d.accumulate = 1
        if variant == "mask_y1":
            d.y1 = hip.view(y1)
        hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), impl), f"impl {impl}")
        torch.cuda.synchronize()
        outs[impl] = (y.float().cpu(), y1.float().cpu())
        ref = _desc_reference(d, st.tensor("c.weight").cpu(), st.tensor("c.bias").cpu(), xb, y_init, r1, None, m)
        _assert_one_ulp(y, ref["y"], (impl, variant, "y"))
        if variant == "mask_y1":
            _assert_one_ulp(y1, ref["y1"], (impl, variant, "y1"))



@pytest.mark.parametrize("variant", ["lrelu", "lrelu_r1", "lrelu_r1_y0", "mask", "mask_acc", "mask_r1", "plain", "generic"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(128, 64, 32, 32, 2), (96, 128, 37, 21, 1), (256, 64, 16, 48, 1)])
def test_big_tile_conv_vs_float64_contract_reference(variant, cin, cout, H, W, B):
    """csrc/conv_big.hip (32x16-pixel x 64-channel workgroup tiles, 4x2 register tiling, rotated lane->pixel map)
    forced through ssr_conv2d_impl(impl=4) against the pipelined kernel (impl=3) on the same descriptor: every
    branch-free epilogue instantiation plus the generic one (alpha, both residuals, dual outputs), ragged sizes."""
    import ctypes as C
    engine, hip = _mods()
    dt, tdt = hip.BF16, torch.bfloat16
    torch.manual_seed(cin + cout + H + len(variant))
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], dt)
    st.load_state_dict({"c.weight": torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5), "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    mk = lambda c: (torch.randn(B, H, W, c, device="cuda") * 0.5).to(tdt).contiguous()
    xb, r1, r2, m, y_init = mk(cin), mk(cout), mk(cout), mk(cout), mk(cout)
    outs = {}
    for impl in (4, 3):
        y, y0 = y_init.clone(), torch.zeros_like(y_init)
        cb = engine._ConvBuilder(st, B)
        L = engine.Launcher()
        kw = dict(act=hip.ACT_LRELU if variant in ("lrelu", "lrelu_r1", "lrelu_r1_y0", "generic") else hip.ACT_NONE, cin=cin)
        if variant in ("mask_r1", "lrelu_r1", "lrelu_r1_y0", "generic"):
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
        hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), impl), f"impl {impl}")
        torch.cuda.synchronize()
        outs[impl] = (y.float().cpu(), y0.float().cpu())
        ref = _desc_reference(d, st.tensor("c.weight").cpu(), st.tensor("c.bias").cpu(), xb, y_init, r1, r2, m)
        _assert_one_ulp(y, ref["y"], (impl, variant, "y"))
        if variant in ("lrelu_r1_y0", "generic"):
            _assert_one_ulp(y0, ref["y0"], (impl, variant, "y0"))


@pytest.mark.parametrize("B,H,W", [(2, 128, 128), (1, 40, 52)])
def test_usm_sharp_matches_oracle(B, H, W):
    """ssr_usm_sharp (separable 51-tap Gaussian in LDS, reflect padding) vs the oracle's 2-D restatement of BasicSR's
    USMSharp on uint8-valued images (feed_data: ssr_esrgan_model.py:108-109).  The residual mask is a threshold, so a
    pixel exactly at |residual|*255 == 10 could flip: allow a handful of outliers, everything else to 2e-5."""
    from oracle import esrgan_oracle as O
    _, hip = _mods()
    torch.manual_seed(B + H)
    base = torch.rand(B, 3, H // 4 + 1, W // 4 + 1)
    img = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear") + 0.08 * torch.randn(B, 3, H, W)
    u8 = (img.clamp(0, 1) * 255).round()
    src = u8.cuda().contiguous()
    dst = torch.empty_like(src)
    hip.check(hip.lib().ssr_usm_sharp(src.data_ptr(), dst.data_ptr(), B * 3, H, W, 1.0 / 255, 0.5, 10.0, hip.stream_ptr()), "usm")
    torch.cuda.synchronize()
    ref = O.usm_sharp(u8 / 255)
    diff = (dst.cpu() - ref).abs()
    assert float((diff > 2e-5).float().mean()) < 1e-3, float(diff.max())
    assert float(diff.max()) < 0.05


def test_train_step_feeds_usm_sharpened_l1_target():
    """l1_gt_usm=True, gan_gt_usm=False (the shipped esrgan_s2naip_urban.yml:9-11): the L1 target is the sharpened
    ground truth, the discriminator's real input is the plain one (ssr_esrgan_model.py:121-129,202-213)."""
    from oracle import esrgan_oracle as O
    from satlas_super_resolution_amd import train_step as T
    torch.manual_seed(3)
    cfg = T.StepConfig(l1_gt_usm=True, gan_gt_usm=False)
    g_kw = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=1, num_grow_ch=32)
    ts = T.ESRGANTrainStep(g_kw, dict(num_in_ch=3, num_feat=64, skip_connection=True), 1, 32, 32, dtype="fp32", cfg=cfg,
                           use_graph=False)
    ts.load_state(O.generator_init(seed=1, **g_kw), O.discriminator_init(3, 64, seed=2))
    lr = torch.randint(0, 256, (1, 3, 32, 32)).float().cuda()
    gt = (torch.rand(1, 3, 33, 33)[..., :32, :32].repeat_interleave(4, 2).repeat_interleave(4, 3) * 255).round().cuda()
    ts.feed_data(lr, gt, scale=1.0 / 255)
    torch.cuda.synchronize()
    assert ts.l1_tgt is not ts.real_in
    l1 = ts.l1_tgt[..., :3].float().cpu().permute(0, 3, 1, 2)
    real = ts.real_in[..., :3].float().cpu().permute(0, 3, 1, 2)
    assert torch.allclose(real, gt.cpu() / 255, atol=1e-6)
    assert float((l1 - O.usm_sharp(gt.cpu() / 255)).abs().max()) < 1e-3
    ts.step(1)
    assert all(math.isfinite(v) for v in ts.log().values())


@pytest.mark.parametrize("variant", ["bias", "lrelu_r1", "mask_acc", "dual"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 1, 16, 40, 2), (64, 3, 9, 33, 1), (24, 3, 8, 32, 1), (64, 8, 16, 32, 1)])
def test_thin_output_conv_vs_float64_contract_reference(variant, cin, cout, H, W, B):
    """csrc/conv_thin.hip (v_dot2c_f32_bf16 dot products, one pixel per thread; the conv9 / conv_last / conv0-dgrad shapes)
    forced through ssr_conv2d_impl(impl=5) against the pipelined MFMA kernel (impl=3), epilogue features included."""
    import ctypes as C
    engine, hip = _mods()
    dt, tdt = hip.BF16, torch.bfloat16
    torch.manual_seed(cin + cout + H + len(variant))
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], dt)
    st.load_state_dict({"c.weight": torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5), "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    cp = 8                                                     # outputs live in 8-channel padded buffers
    mk = lambda c: (torch.randn(B, H, W, c, device="cuda") * 0.5).to(tdt).contiguous()
    xb, r1, m, y_init = mk(cin), mk(cp), mk(cp), mk(cp)
    outs = {}
    for impl in (5, 3):
        y, y0, y1 = y_init.clone(), torch.zeros_like(y_init), torch.zeros_like(y_init)
        cb = engine._ConvBuilder(st, B)
        L = engine.Launcher()
        kw = dict(act=hip.ACT_LRELU if variant in ("lrelu_r1", "dual") else hip.ACT_NONE, cin=cin)
        if variant in ("lrelu_r1", "dual"):
            kw.update(r1=hip.view(r1), r1_nc=cout, beta1=0.5)
        if variant == "dual":
            kw.update(alpha=0.7, y0=hip.view(y0))
        d = cb.conv(L, "c", hip.view(xb), H, W, hip.view(y), **kw)
        if variant in ("mask_acc", "dual"):
            d.m, d.m_c0, d.m_c1 = hip.view(m), 0, cout
        if variant == "mask_acc":
            d.accumulate = 1
        if variant == "dual":
            d.y1 = hip.view(y1)
        hip.check(hip.lib().ssr_conv2d_impl(C.byref(d), hip.stream_ptr(), impl), f"impl {impl}")
        torch.cuda.synchronize()
        outs[impl] = tuple(t[..., :cout].float().cpu() for t in (y, y0, y1))
        ref = _desc_reference(d, st.tensor("c.weight").cpu(), st.tensor("c.bias").cpu(), xb, y_init, r1, None, m)
        _assert_one_ulp(y, ref["y"], (impl, variant, "y"))
        if variant == "dual":
            _assert_one_ulp(y0, ref["y0"], (impl, variant, "y0"))
            _assert_one_ulp(y1, ref["y1"], (impl, variant, "y1"))


@pytest.mark.parametrize("cout,cin,gh,gw,B", [(128, 64, 32, 32, 2), (64, 64, 20, 24, 1), (256, 128, 16, 16, 1)])
def test_stride2_dgrad_big_tile_parity_classes(cout, cin, gh, gw, B, monkeypatch):
    """dgrad of a 4x4 stride-2 conv (discriminator_arch.py:31-33 conv1..conv3) = four 2x2 parity-class convs.  The
    big-tile 2x2 instantiation of csrc/conv_big.hip — each class forced through impl=4, and all four in ONE launch through
    ssr_conv2d_batch — against the pipelined kernel (impl=3) on the same descriptors, with the step's epilogue
    (residual + LeakyReLU-backward mask)."""
    import ctypes as C
    engine, hip = _mods()
    dt, tdt = hip.BF16, torch.bfloat16
    torch.manual_seed(cout + cin + gh)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 4, 2, False, True)], dt)
    st.load_state_dict({"c.weight_orig": torch.randn(cout, cin, 4, 4) * (1.0 / (cin * 16) ** 0.5),
                        "c.weight_u": torch.randn(cout), "c.weight_v": torch.randn(cin * 16)})
    st.spectral_norm(power_iter=False)
    st.pack()
    mk = lambda h, w, c: (torch.randn(B, h, w, c, device="cuda") * 0.5).to(tdt).contiguous()
    dy, r1, m = mk(gh, gw, cout), mk(2 * gh, 2 * gw, cin), mk(2 * gh, 2 * gw, cin)
    outs = {}
    for mode in ("pipelined", "big_each", "big_batch"):
        y = torch.zeros(B, 2 * gh, 2 * gw, cin, device="cuda", dtype=tdt)
        cb = engine._ConvBuilder(st, B)
        L = engine.Launcher()
        cb.dgrad(L, "c", hip.view(dy), gh, gw, hip.view(y), r1=hip.view(r1), r1_nc=cin, beta1=1.0, m=hip.view(m), m_c0=0, m_c1=cin)
        fn, args, _ = L.calls[0]
        arr, n = args[0], args[1]
        if mode == "big_batch":
            monkeypatch.setenv("SSR_CONV_BIGTILE2", "2")
            hip.check(hip.lib().ssr_conv2d_batch(arr, n, hip.stream_ptr()), "batch")
            monkeypatch.delenv("SSR_CONV_BIGTILE2")
        else:
            for k in range(n):
                hip.check(hip.lib().ssr_conv2d_impl(C.byref(arr[k]), hip.stream_ptr(), 3 if mode == "pipelined" else 4), mode)
        torch.cuda.synchronize()
        outs[mode] = y.float().cpu()
    assert rel_err(outs["big_each"], outs["pipelined"]) < 1e-2
    assert rel_err(outs["big_batch"], outs["pipelined"]) < 1e-2


@pytest.mark.parametrize("cin,cout,H,W,B", [(64, 128, 32, 32, 2), (128, 256, 24, 40, 1), (256, 512, 16, 16, 2), (32, 64, 66, 34, 1)])
def test_stride2_forward_space_to_depth(cin, cout, H, W, B):
    """4x4 stride-2 spectral-norm conv + LeakyReLU (discriminator_arch.py:31-33,45-47) through ssr_conv_desc.s2d — a 2x2
    conv over a space-to-depth view gathered by the big-tile kernel's staging loads, weights packed in that order
    (ssr_pack_item.fwd_s2d) — against torch on the bf16-rounded operands."""
    engine, hip = _mods()
    dt, tdt = hip.BF16, torch.bfloat16
    torch.manual_seed(cin + cout + H)
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 4, 2, False, True)], dt)
    assert st.s2d["c"], "shape should qualify for the space-to-depth path"
    w = torch.randn(cout, cin, 4, 4) * (1.0 / (cin * 16) ** 0.5)
    st.load_state_dict({"c.weight_orig": w, "c.weight_u": torch.randn(cout), "c.weight_v": torch.randn(cin * 16)})
    st.spectral_norm(power_iter=False)
    st.pack()
    sigma = float(st.sigma[0])
    x = torch.randn(B, cin, H, W)
    xb = torch.zeros(B, H, W, cin, dtype=tdt, device="cuda")
    yb = torch.zeros(B, H // 2, W // 2, cout, dtype=tdt, device="cuda")
    _nchw_to_buf(hip, x, xb, dt)
    cb = engine._ConvBuilder(st, B)
    L = engine.Launcher()
    d = cb.conv(L, "c", hip.view(xb), H, W, hip.view(yb), act=hip.ACT_LRELU)
    assert d.s2d == 1 and hip.lib().ssr_conv2d_variant(d) % 10 == 9
    L.run()
    y = _buf_to_nchw(hip, yb, cout, dt).cpu()
    yr = F.leaky_relu(F.conv2d(x.bfloat16().float(), (w / sigma).bfloat16().float(), None, stride=2, padding=1), 0.2)
    assert rel_err(y, yr) < 1.5e-2, rel_err(y, yr)


@pytest.mark.parametrize("groups", ["1", "2"])
def test_big_tile_workgroups_persistent_over_images(groups, monkeypatch):
    """The big-tile kernels walk images n0, n0 + G, ... per workgroup, the chunk stream crossing image boundaries
    (csrc/conv_big.hip).  G is chosen from the grid size, which the small parity shapes never push below N; the
    SSR_CONV_BIG_G hook forces it: B = 3 with G = 1 (one workgroup per tile position, three images) and G = 2 (uneven: two
    images / one image), for the in-stream and the final epilogue of every family — 3x3 lean and generic, the batched
    2x2 parity classes of the stride-2 dgrad and the space-to-depth forward."""
    monkeypatch.setenv("SSR_CONV_BIG_G", groups)
    for variant in ("plain", "lrelu", "lrelu_r1_y0", "mask_acc", "mask_r1", "generic"):
        test_big_tile_conv_vs_float64_contract_reference(variant, 128, 64, 32, 32, 3)
    test_big_tile_conv_vs_float64_contract_reference("lrelu", 32, 64, 37, 21, 3)        # a single chunk per image
    test_stride2_dgrad_big_tile_parity_classes(128, 64, 32, 32, 3, monkeypatch)
    test_stride2_forward_space_to_depth(64, 128, 32, 32, 3)
    test_stride2_forward_space_to_depth(32, 64, 66, 34, 3)                         # nchunks = 4 (one per parity class)
