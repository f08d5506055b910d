"""Diagnostic: per-parameter gradient error of the generator plan vs the CPU oracle (fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satlas_super_resolution_amd import engine, hip
from oracle import esrgan_oracle as O
torch.manual_seed(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kw = dict(num_in_ch=24, num_out_ch=3, scale=4, num_feat=64, num_block=nb, num_grow_ch=32)
sd = O.generator_init(seed=0, **kw)
x = torch.rand(2, 24, 32, 32)
st = engine.ParamStore(engine.generator_specs(**kw), hip.F32)
st.load_state_dict(sd)
plan = engine.GeneratorPlan(st, 2, 32, 32, training=True, need_input_grad=True, **kw)
st.pack(); plan.load_input(x.cuda()); plan.fwd.run()
y = plan.read_output().cpu()
sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
xg = x.clone().requires_grad_(True)
ref = O.generator_forward(sdg, xg)
g = torch.randn_like(ref)
(ref * g).sum().backward()
plan.load_output_grad(g.cuda()); st.grad.zero_(); plan.bwd.run(); torch.cuda.synchronize()
print("fwd err", float((y - ref.detach()).abs().max() / ref.abs().max()))
errs = []
for k in sd:
    gr = sdg[k].grad
    got = st.tensor(k, st.grad).cpu()
    e = float((got - gr).abs().max() / gr.abs().max())
    errs.append((e, k, float(gr.abs().max())))
for e, k, m in errs:
    if e > 2e-4 or k.startswith("conv_"):
        print(f"{e:.3e} {k} max|g|={m:.3e}")
dx = plan.read_input_grad().cpu()
print("dx err", float((dx - xg.grad).abs().max() / xg.grad.abs().max()))
# 64-bit reference to separate oracle rounding from ours
sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
ref64 = O.generator_forward(sd64, x.double())
(ref64 * g.double()).sum().backward()
for k in ("conv_first.weight", "conv_first.bias", "body.0.rdb1.conv1.weight", "conv_last.weight"):
    g64 = sd64[k].grad
    e_ours = float((st.tensor(k, st.grad).cpu().double() - g64).abs().max() / g64.abs().max())
    e_orc = float((sdg[k].grad.double() - g64).abs().max() / g64.abs().max())
    print(f"{k}: ours-vs-fp64 {e_ours:.3e}   oracle(fp32 CPU)-vs-fp64 {e_orc:.3e}")
