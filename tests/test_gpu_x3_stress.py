"""GPU stress test of the LDS hand-over protocols of the fp32x3 mode: the register-tiled kernel of the generator body (round 6,
csrc/conv_x3r.hip: `pdone[producer]` / `cdone[wave]` counters, four producer + four MFMA waves over an eight-stage patch ring, no
barrier in the loop), the twelve-wave ring kernel it replaced (round 5, csrc/conv_x3q.hip: kept behind SSR_X3_REGTILE=0) and the
one-pass weight-gradient kernel (csrc/wgrad_x3.hip: loader waves splitting fp32 tiles into a two-stage ring).  They replace
nn.Conv2d forward / dgrad / weight gradients of /root/reference/ssr/archs/rrdbnet_arch.py:26-44.

Both kernels are deterministic for a fixed launch (one writer per output element), so every launch must reproduce the FIRST launch's
bytes, which the parity tests (tests/test_gpu_conv_x3.py, tests/test_gpu_wgrad_x3.py) pin against float64 torch; thousands of launches
at the benchmarked launch shape (B = 32, 32 x 32 pixels: 256 workgroups of twelve waves), alone and while a second stream keeps the
CUs / L2 / HBM busy.  Why it exists: DESIGN.md lessons 36 / 38 - a hand-over bug in the bf16 dense block showed up as a few wrong
tiles in one launch of thousands, under load only; one launch per shape cannot see that."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
# default: 1000 launches per case (the driver's GPU-test budget is 1200 s for the whole suite); SSR_STRESS_LAUNCHES_X3=4000 is the
# round-5 depth, run once per round by tools/gpu_round.sh
LAUNCHES = int(os.environ.get("SSR_STRESS_LAUNCHES_X3", "1000"))
B, H, W = 32, 32, 32


def _side_load(side):
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    bm = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    big = torch.empty(64 * 1024 * 1024, device="cuda", dtype=torch.float32)
    big2 = torch.empty_like(big)
    state = {"k": 0}

    def more():
        with torch.cuda.stream(side):
            if state["k"] & 1:
                big2.copy_(big, non_blocking=True)
            else:
                torch.mm(a, bm)
        state["k"] += 1
    return more


@pytest.mark.parametrize("loaded", [False, True], ids=["alone", "second_stream_busy"])
@pytest.mark.parametrize("impl", [0, 6], ids=["regtile", "ring"])
@pytest.mark.parametrize("cin,cout", [(160, 32), (192, 64)], ids=["nt1_conv4", "nt2_conv5"])
def test_thousands_of_body_kernel_launches_reproduce_the_first(cin, cout, loaded, impl):
    """dense-block conv4 (10 chunks: the eight-stage ring of the register-tiled kernel wraps, producers wait for the MFMA waves) and
    conv5 (12 chunks, 64 output channels: the producer waves finish the second channel tile); impl 6 = the round-5 ring kernel"""
    n_launch = LAUNCHES if impl == 0 else min(LAUNCHES, 300)       # the replaced kernel: a short run (suite time)
    from satlas_super_resolution_amd import engine, hip
    lib = hip.lib()
    torch.manual_seed(cin + cout + int(loaded))
    st = engine.ParamStore([engine.ConvSpec("c", cout, cin, 3, 1, True, False)], hip.F32X3)
    st.load_state_dict({"c.weight": torch.randn(cout, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5), "c.bias": torch.randn(cout) * 0.1})
    st.pack()
    buf = (torch.randn(B, H, W, 192, device="cuda") * 0.5).contiguous()      # the block buffer: the conv reads a channel prefix
    NSET = 3
    outs = [torch.zeros(B, H, W, cout, device="cuda") for _ in range(NSET)]
    cb = engine._ConvBuilder(st, B)
    descs = [cb.conv(engine.Launcher(), "c", hip.view(buf, 0), H, W, hip.view(outs[k]), act=hip.ACT_LRELU, cin=cin) for k in range(NSET)]
    assert lib.ssr_conv2d_variant(C.byref(descs[0])) % 10 == 5              # the register-tiled kernel is what ssr_conv2d runs here
    ref = torch.zeros_like(outs[0])
    dref = cb.conv(engine.Launcher(), "c", hip.view(buf, 0), H, W, hip.view(ref), act=hip.ACT_LRELU, cin=cin)
    assert lib.ssr_conv2d_impl(C.byref(dref), None, impl) == 0
    torch.cuda.synchronize()
    assert float(ref.abs().max()) > 0.1
    ref_i = ref.view(torch.int32)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    more = _side_load(side) if loaded else None
    bad = torch.zeros(2, device="cuda", dtype=torch.int64)
    sp = main.cuda_stream
    for it in range(n_launch):
        k = it % NSET
        if loaded and it % 8 == 0:
            more()
        outs[k].zero_()
        assert lib.ssr_conv2d_impl(C.byref(descs[k]), sp, impl) == 0
        bad[0] += (outs[k].view(torch.int32) != ref_i).sum()
        bad[1] += 1
    torch.cuda.synchronize()
    nb = bad.cpu().tolist()
    assert nb[1] == n_launch and nb[0] == 0, f"{nb[0]} differing words in {n_launch} launches"


@pytest.mark.parametrize("loaded", [False, True], ids=["alone", "second_stream_busy"])
def test_hundreds_of_one_pass_weight_gradient_launches_reproduce_the_first(loaded):
    """a dense block's conv1..conv4 over one buffer (paired items: two layers per workgroup) at B = 32; one pixel range per item, so
    every gradient element has one writer and the launch is deterministic"""
    from satlas_super_resolution_amd import engine, hip
    n = max(50, LAUNCHES // 5)
    torch.manual_seed(77 + int(loaded))
    x = (torch.randn(B, H, W, 192, device="cuda") * 0.5).contiguous()
    layers = [(64, 32), (96, 32), (128, 32), (160, 32)]
    wb = engine.WgradBatch(hip.F32X3, 3, 1)
    assert wb.kdt == hip.F32X3
    wb.MAX_TILES_PER_ITEM = {3: 1 << 20}
    dys, dws, dbs = [], [], []
    for cin, cout in layers:
        dy = (torch.randn(B, H, W, cout, device="cuda") * 0.25).contiguous()
        dw, db = torch.zeros(cout, cin, 3, 3, device="cuda"), torch.zeros(cout, device="cuda")
        wb.add(hip.view(x), hip.view(dy), B, H, W, 1, cin, cout, H, W, 1.0, dw.data_ptr(), cin, db.data_ptr())
        dys.append(dy); dws.append(dw); dbs.append(db)
    wb.finalize()
    assert all(it.atomic == 0 for it in wb.items) and any(it.nco == 2 for it in wb.items)
    L = engine.Launcher()
    wb.launch(L)
    L.run()
    torch.cuda.synchronize()
    ref_w, ref_b = [t.clone().view(torch.int32) for t in dws], [t.clone().view(torch.int32) for t in dbs]
    assert all(float(t.abs().max()) > 0.1 for t in dws)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    more = _side_load(side) if loaded else None
    bad = torch.zeros(2, device="cuda", dtype=torch.int64)
    for it in range(n):
        if loaded and it % 2 == 0:
            more()
        for t in dws + dbs:
            t.zero_()
        L.run()
        for t, r in zip(dws + dbs, ref_w + ref_b):
            bad[0] += (t.view(torch.int32) != r).sum()
        bad[1] += 1
    torch.cuda.synchronize()
    nb = bad.cpu().tolist()
    assert nb[1] == n and nb[0] == 0, f"{nb[0]} differing words in {n} launches"


@pytest.mark.parametrize("loaded", [False, True], ids=["alone", "second_stream_busy"])
def test_hundreds_of_chain_launches_reproduce_four_launches(loaded):
    """the persistent dense-block chain (csrc/conv_x3c.hip, opt-in SSR_X3_CHAIN=1): conv1..conv4 in one launch of 256 workgroups that hand
    their results to each other through memory (write-through stores, per-tile flag bytes, ticketed tiles, self re-arming epoch).  Every
    launch must give the bytes of four register-tiled launches - also while a second stream takes CUs away (workgroups of an image
    then start at different times: the case the ticket order exists for)."""
    from satlas_super_resolution_amd import engine, hip
    lib = hip.lib()
    nf, gc = 64, 32
    torch.manual_seed(3 + int(loaded))
    specs = [engine.ConvSpec(f"c{k}", gc, nf + (k - 1) * gc, 3, 1, True, False) for k in range(1, 5)]
    st = engine.ParamStore(specs, hip.F32X3)
    sd = {}
    for k in range(1, 5):
        cin = nf + (k - 1) * gc
        sd[f"c{k}.weight"] = torch.randn(gc, cin, 3, 3) * (1.0 / (cin * 9) ** 0.5)
        sd[f"c{k}.bias"] = torch.randn(gc) * 0.1
    st.load_state_dict(sd)
    st.pack()
    x0 = (torch.randn(B, H, W, nf, device="cuda") * 0.5).contiguous()
    state = torch.zeros(int(lib.ssr_conv2d_chain_state_bytes(B, H, W) + 3) // 4, dtype=torch.int32, device="cuda")
    cb = engine._ConvBuilder(st, B)

    def descs_on(buf):
        ds = [cb.conv(engine.Launcher(), f"c{k}", hip.view(buf, 0), H, W, hip.view(buf, nf + (k - 1) * gc), act=hip.ACT_LRELU, cin=nf + (k - 1) * gc)
              for k in range(1, 5)]
        arr = (hip.ConvDesc * 4)()
        for i, d in enumerate(ds):
            C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(hip.ConvDesc))
        return arr

    ref = torch.zeros(B, H, W, nf + 4 * gc, device="cuda")
    ref[..., :nf] = x0
    aref = descs_on(ref)
    for k in range(4):
        assert lib.ssr_conv2d_impl(C.byref(aref[k]), None, 7) == 0
    torch.cuda.synchronize()
    ref_i = ref.view(torch.int32)
    NSET = 3
    bufs = [torch.zeros_like(ref) for _ in range(NSET)]
    arrs = [descs_on(b) for b in bufs]
    assert lib.ssr_conv2d_chain_ok(arrs[0], 4) == 1
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    more = _side_load(side) if loaded else None
    bad = torch.zeros(2, device="cuda", dtype=torch.int64)
    n = max(100, LAUNCHES // 4)
    for it in range(n):
        k = it % NSET
        if loaded and it % 4 == 0:
            more()
        bufs[k].zero_()
        bufs[k][..., :nf] = x0
        assert lib.ssr_conv2d_chain(arrs[k], 4, state.data_ptr(), main.cuda_stream) == 0
        bad[0] += (bufs[k].view(torch.int32) != ref_i).sum()
        bad[1] += 1
    torch.cuda.synchronize()
    nb = bad.cpu().tolist()
    assert nb[1] == n and nb[0] == 0, f"{nb[0]} differing words in {n} chain launches"
