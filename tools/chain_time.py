"""launch time of a dense block's conv1..conv4 (fp32x3): four launches of the register-tiled kernel against ONE persistent chain launch
(csrc/conv_x3c.hip), forward and gather-form backward, back to back on one stream:  python tools/chain_time.py [B=32] [reps=100]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satlas_super_resolution_amd import engine, hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
H = W = 32
nf, gc = 64, 32
lib = hip.lib()
names = [f"body.0.rdb1.conv{k}" for k in range(1, 6)]
specs = [engine.ConvSpec(names[k - 1], gc if k < 5 else nf, nf + (k - 1) * gc, 3, 1, True, False, dgrad_packed=False) for k in range(1, 6)]
st = engine.ParamStore(specs, hip.F32X3)
torch.manual_seed(0)
st.load_state_dict({**{n + ".weight": torch.randn(s.cout, s.cin, 3, 3) * 0.02 for n, s in zip(names, specs)},
                    **{n + ".bias": torch.zeros(s.cout) for n, s in zip(names, specs)}})
st.add_rdb_gather("body.0.rdb1", nf, gc, 0.2)
st.pack()
NB = 6      # distinct blocks in rotation (as the step's 69: no launch re-reads what the previous one just wrote)
curs = [(torch.randn(B, H, W, nf + 4 * gc, device="cuda") * 0.5).contiguous() for _ in range(NB)]
dcurs = [torch.zeros(B, H, W, nf + 4 * gc, device="cuda") for _ in range(NB)]
d_out = (torch.randn(B, H, W, nf, device="cuda") * 0.5).contiguous()
state = torch.zeros(int(lib.ssr_conv2d_chain_state_bytes(B, H, W) + 3) // 4, dtype=torch.int32, device="cuda")
cb = engine._ConvBuilder(st, B)


def arr_of(ds):
    arr = (hip.ConvDesc * len(ds))()
    for i, d in enumerate(ds):
        C.memmove(C.byref(arr[i]), C.byref(d), C.sizeof(hip.ConvDesc))
    return arr


fwd = [arr_of([cb.conv(engine.Launcher(), names[k - 1], hip.view(c, 0), H, W, hip.view(c, nf + (k - 1) * gc), act=hip.ACT_LRELU, cin=nf + (k - 1) * gc)
               for k in range(1, 5)]) for c in curs]
bwd = [arr_of([engine.gather_dgrad(cb, engine.Launcher(), "body.0.rdb1", k, hip.view(dc, nf + k * gc), (4 - k) * gc, hip.view(d_out), nf, H, W,
                                   hip.view(dc, nf + (k - 1) * gc), gc, m=hip.view(c, nf + (k - 1) * gc), m_c0=0, m_c1=gc) for k in (4, 3, 2, 1)])
       for c, dc in zip(curs, dcurs)]
sp = hip.stream_ptr()


def run(arrs, chain):
    def one(a):
        if chain:
            assert lib.ssr_conv2d_chain(a, 4, state.data_ptr(), sp) == 0
        else:
            for k in range(4):
                assert lib.ssr_conv2d(C.byref(a[k]), sp) == 0
    for a in arrs:
        one(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(REPS):
        one(arrs[r % NB])
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / REPS


assert lib.ssr_conv2d_chain_ok(fwd[0], 4) == 1 and lib.ssr_conv2d_chain_ok(bwd[0], 4) == 1
for name, arrs in (("forward conv1-4", fwd), ("backward slices 4-1", bwd)):
    for rep in range(2):
        t4, t1 = run(arrs, False), run(arrs, True)
        print(f"B={B} {name}: four launches {t4:7.2f} us | one chain launch {t1:7.2f} us  ({t1 / t4:.3f}x)")
