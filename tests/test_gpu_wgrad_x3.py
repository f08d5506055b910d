"""GPU parity of the one-pass fp32x3 weight gradient of the 3x3 stride-1 layers, csrc/wgrad_x3.hip: loader waves split fp32 tiles
into bf16 hi / lo planes on the way into LDS, three MFMAs per fragment pair into one accumulator set, paired items (two blocks of
32 output-gradient channels against one 64-channel input patch), 8 x 16-pixel tiles.

Every case is checked against autograd of nn.functional.conv2d in float64 (the layers are the nn.Conv2d 3x3 of
/root/reference/ssr/archs/rrdbnet_arch.py:30-34,104-113 and discriminator_arch.py:28-40; their weight / bias gradients come from
l_g_total.backward() / l_d_real.backward() / l_d_fake.backward(), ssr/models/ssr_esrgan_model.py:188,219,227) at 1e-4 of max|ref|
(a product rounds at 2^-16: ~1e-5 per layer), and against the older three-pass form (SSR_X3_WGRAD_FUSED=0) at the same level."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _mods():
    from satlas_super_resolution_amd import engine, hip
    return engine, hip


def _ref_grads(x, dys, up):
    """x: [B, H, W, C] fp32 (device layout), dys: list of (dy [B, Ho, Wo, Cout_pad], cin, cout, alpha) -> float64 (dW, db) per layer"""
    out = []
    for dy, cin, cout, alpha in dys:
        xi = x.double().cpu().permute(0, 3, 1, 2)[:, :cin]
        if up == 2:
            xi = F.interpolate(xi, scale_factor=2, mode="nearest")
        w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
        b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xi, w, b, padding=1)
        g = dy.double().cpu().permute(0, 3, 1, 2)[:, :cout]
        (y * g).sum().backward()
        out.append((alpha * w.grad, alpha * b.grad))
    return out


def _run(layers, B, H, W, up=1, cbuf=None, max_tiles=None, det=False, fused=True, with_bias=True):
    """layers: list of (cin, cout, alpha): every layer reads channels [0, cin) of ONE input buffer (a dense block's convs do)."""
    engine, hip = _mods()
    old = os.environ.get("SSR_X3_WGRAD_FUSED")
    os.environ["SSR_X3_WGRAD_FUSED"] = "1" if fused else "0"
    try:
        torch.manual_seed(B * 1000 + H * 10 + len(layers))
        cbuf = cbuf or engine.rup(max(c for c, _, _ in layers), 8)
        Ho, Wo = H * up, W * up
        x = (torch.randn(B, H, W, cbuf, device="cuda") * 0.5).contiguous()
        wb = engine.WgradBatch(hip.F32X3, 3, 1, det=det)
        assert (wb.kdt == hip.F32X3) == fused
        if max_tiles:
            wb.MAX_TILES_PER_ITEM = {3: max_tiles}
        dys, grads = [], []
        for cin, cout, alpha in layers:
            coutp = engine.rup(cout, 8)
            dy = (torch.randn(B, Ho, Wo, coutp, device="cuda") * 0.25).contiguous()
            dy[..., cout:] = 0
            dw = torch.zeros(cout, cin, 3, 3, device="cuda")
            db = torch.zeros(cout, device="cuda")
            wb.add(hip.view(x), hip.view(dy), B, H, W, up, engine.rup(cin, 8), cout, Ho, Wo, alpha, dw.data_ptr(), cin,
                   db.data_ptr() if with_bias else None)
            dys.append((dy, cin, cout, alpha))
            grads.append((dw, db))
        wb.finalize()
        L = engine.Launcher()
        wb.launch(L)
        L.run()
        torch.cuda.synchronize()
        return wb, [(dw.double().cpu(), db.double().cpu()) for dw, db in grads], _ref_grads(x, dys, up)
    finally:
        if old is None:
            os.environ.pop("SSR_X3_WGRAD_FUSED", None)
        else:
            os.environ["SSR_X3_WGRAD_FUSED"] = old


def _check(got, ref, with_bias=True):
    for k, ((dw, db), (rw, rb)) in enumerate(zip(got, ref)):
        assert rel_err(dw, rw) < TOL, ("dW", k, rel_err(dw, rw))
        if with_bias:
            assert rel_err(db, rb) < TOL, ("db", k, rel_err(db, rb))


# a dense block's five convs over one 192-channel buffer (rrdbnet_arch.py:37-41): conv1..4 pair up, conv5's halves pair with each other
DENSE = [(64, 32, 1.0), (96, 32, 1.0), (128, 32, 0.5), (160, 32, 1.0), (192, 64, 0.2)]


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 16, 48), (3, 13, 21)])
def test_dense_block_layers_paired(B, H, W):
    wb, got, ref = _run(DENSE, B, H, W, cbuf=192)
    assert any(it.nco == 2 and it.layer != it.layer_b for it in wb.items), "two layers share an item"
    assert any(it.nco == 2 and it.layer == it.layer_b for it in wb.items), "the halves of the 64-output conv share an item"
    _check(got, ref)


@pytest.mark.parametrize("layers,B,H,W,up", [
    ([(64, 64, 1.0)], 2, 16, 16, 2),            # conv_up1 / conv_up2: nearest x2 folded into the read (rrdbnet_arch.py:127-130)
    ([(3, 64, 1.0)], 2, 24, 40, 1),             # discriminator conv0: 3 input channels in an 8-channel buffer
    ([(64, 3, 1.0)], 2, 24, 40, 1),             # conv_last / conv9-like thin outputs
    ([(64, 1, 1.0)], 1, 9, 7, 1),
    ([(40, 8, 1.0), (24, 16, 1.0)], 3, 5, 33, 1),
    ([(512, 256, 1.0)], 1, 16, 16, 1),          # discriminator conv4-like: many (co, ci) tiles
    ([(128, 64, 1.0), (128, 64, 1.0)], 1, 64, 64, 1),
])
def test_layer_shapes(layers, B, H, W, up):
    wb, got, ref = _run(layers, B, H, W, up=up)
    _check(got, ref)


def test_pixel_splits_accumulate_atomically_and_without_bias():
    wb, got, ref = _run(DENSE[:3], 4, 32, 32, cbuf=128, max_tiles=3)
    assert max(it.atomic for it in wb.items) == 1 and len(wb.items) > 20
    _check(got, ref)
    wb, got, ref = _run([(64, 32, 1.0)], 2, 16, 16, with_bias=False)
    _check(got, ref, with_bias=False)
    assert all(float(db.abs().max()) == 0.0 for _, db in got)


def test_deterministic_mode_is_bit_reproducible():
    a = _run(DENSE[:4], 4, 32, 32, cbuf=160, max_tiles=4, det=True)
    b = _run(DENSE[:4], 4, 32, 32, cbuf=160, max_tiles=4, det=True)
    assert a[0].partial is not None
    _check(a[1], a[2])
    for (dw1, db1), (dw2, db2) in zip(a[1], b[1]):
        assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


def test_agrees_with_three_pass_form():
    _, fused, ref = _run(DENSE, 2, 32, 32, cbuf=192)
    _, three, _ = _run(DENSE, 2, 32, 32, cbuf=192, fused=False)
    for (dw1, db1), (dw2, db2) in zip(fused, three):
        assert rel_err(dw1, dw2) < TOL and rel_err(db1, db2) < TOL


# ---- 4x4 stride 2 (round 6, wgrad_x3_k4_kernel): the discriminator's conv1..conv3 (discriminator_arch.py:31-33) ----
def _run4(layers, B, H, W, max_tiles=None, det=False, fused=True, with_bias=False):
    """layers: list of (cin, cout, alpha), each with its own input buffer [B, H, W, cin] and output-gradient buffer [B, H/2, W/2, cout]"""
    engine, hip = _mods()
    old = os.environ.get("SSR_X3_WGRAD_FUSED4")
    os.environ["SSR_X3_WGRAD_FUSED4"] = "1" if fused else "0"
    try:
        torch.manual_seed(B * 1000 + H * 10 + len(layers) + 4)
        wb = engine.WgradBatch(hip.F32X3, 4, 2, det=det)
        assert (wb.kdt == hip.F32X3) == fused
        if max_tiles:
            wb.MAX_TILES_PER_ITEM = {4: max_tiles}
        Ho, Wo = (H + 2 - 4) // 2 + 1, (W + 2 - 4) // 2 + 1
        keep, got, ref = [], [], []
        for cin, cout, alpha in layers:
            cinp, coutp = engine.rup(cin, 8), engine.rup(cout, 8)
            x = (torch.randn(B, H, W, cinp, device="cuda") * 0.5).contiguous()
            x[..., cin:] = 0
            dy = (torch.randn(B, Ho, Wo, coutp, device="cuda") * 0.25).contiguous()
            dy[..., cout:] = 0
            dw = torch.zeros(cout, cin, 4, 4, device="cuda")
            db = torch.zeros(cout, device="cuda")
            wb.add(hip.view(x), hip.view(dy), B, H, W, 1, cinp, cout, Ho, Wo, alpha, dw.data_ptr(), cin, db.data_ptr() if with_bias else None)
            keep.append((x, dy, dw, db))
            xi = x.double().cpu().permute(0, 3, 1, 2)[:, :cin]
            w = torch.zeros(cout, cin, 4, 4, dtype=torch.float64, requires_grad=True)
            b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
            (F.conv2d(xi, w, b, stride=2, padding=1) * dy.double().cpu().permute(0, 3, 1, 2)[:, :cout]).sum().backward()
            ref.append((alpha * w.grad, alpha * b.grad))
        wb.finalize()
        L = engine.Launcher()
        wb.launch(L)
        L.run()
        torch.cuda.synchronize()
        return wb, [(dw.double().cpu(), db.double().cpu()) for _, _, dw, db in keep], ref
    finally:
        if old is None:
            os.environ.pop("SSR_X3_WGRAD_FUSED4", None)
        else:
            os.environ["SSR_X3_WGRAD_FUSED4"] = old


@pytest.mark.parametrize("layers,B,H,W", [
    ([(64, 128, 1.0)], 2, 32, 32),               # conv1's shape class
    ([(128, 256, 0.5), (64, 128, 1.0)], 1, 24, 40),   # two layers of different grids cannot share a batch: same grid here, ragged tiles (12 x 20)
    ([(256, 512, 1.0)], 2, 16, 16),              # conv3: many (co, ci) tiles, a single 4-row tile band per image row pair
    ([(16, 24, 1.0)], 3, 10, 14),                # half-filled channel tiles, 5 x 7 outputs
    ([(40, 72, 1.0)], 1, 66, 34),                # 33 x 17 outputs: ragged in both directions, three output-channel blocks
])
def test_stride2_4x4_layers_one_pass(layers, B, H, W):
    wb, got, ref = _run4(layers, B, H, W)
    _check(got, ref, with_bias=False)


def test_stride2_4x4_pixel_splits_bias_and_deterministic_mode():
    wb, got, ref = _run4([(64, 128, 1.0)], 4, 32, 32, max_tiles=3, with_bias=True)
    assert max(it.atomic for it in wb.items) == 1 and len(wb.items) > 8 and any(it.nco == 2 for it in wb.items)      # paired blocks of one layer
    _check(got, ref, with_bias=True)
    a = _run4([(64, 64, 1.0)], 4, 32, 32, max_tiles=4, det=True, with_bias=True)
    b = _run4([(64, 64, 1.0)], 4, 32, 32, max_tiles=4, det=True, with_bias=True)
    assert a[0].partial is not None
    _check(a[1], a[2], with_bias=True)
    for (dw1, db1), (dw2, db2) in zip(a[1], b[1]):
        assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


def test_stride2_4x4_single_items_when_pairing_is_off(monkeypatch):
    """SSR_WGRAD_PAIR=0: one 32-channel block of dY per item (the kernel's single form), and a layer with an odd number of blocks keeps a single"""
    monkeypatch.setenv("SSR_WGRAD_PAIR", "0")
    wb, got, ref = _run4([(64, 128, 1.0)], 2, 32, 32, with_bias=True)
    assert all(it.nco == 1 for it in wb.items)
    _check(got, ref, with_bias=True)
    monkeypatch.delenv("SSR_WGRAD_PAIR")
    wb, got, ref = _run4([(40, 72, 1.0)], 1, 66, 34)             # three output-channel blocks: one pair + one single per input tile
    assert any(it.nco == 2 for it in wb.items) and any(it.nco == 1 for it in wb.items)
    _check(got, ref, with_bias=False)


def test_stride2_4x4_agrees_with_three_pass_form():
    _, fused, ref = _run4([(128, 256, 1.0)], 2, 32, 32)
    _, three, _ = _run4([(128, 256, 1.0)], 2, 32, 32, fused=False)
    for (dw1, _), (dw2, _) in zip(fused, three):
        assert rel_err(dw1, dw2) < TOL
