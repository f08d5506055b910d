"""Generator inference throughput through the drop-in nn.Module (SSR_RRDBNet under torch.no_grad, bf16 compute), the path
the reference's infer.py / infer_grid.py call per chunk (ssr/infer_grid.py:97-110): 8xS2 (24-ch) 32x32 -> 128x128 chunks.
Prints images/s with inputs resident in HBM, for the module call (NCHW fp32 in/out, layout kernels included) and for
the bare launch list (plan.fwd.run()).  Supplementary number; the round's headline metric is bench.py's train step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satlas_super_resolution_amd.archs.rrdbnet_arch import SSR_RRDBNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = SSR_RRDBNet(24, 3, 4, 64, 23, 32, compute_dtype="bf16").cuda().eval()
x = torch.rand(B, 24, 32, 32, device="cuda")
with torch.no_grad():
    for _ in range(3):
        y = net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        y = net(x)
    torch.cuda.synchronize()
    t_mod = (time.perf_counter() - t0) / 20
    plan = net.plan(B, 32, 32, training=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.fwd.run()
    torch.cuda.synchronize()
    t_run = (time.perf_counter() - t0) / 20
assert y.shape == (B, 3, 128, 128) and bool(torch.isfinite(y).all())
import json
gflop = 36.739   # algorithmic GFLOP per 32x32 chunk, 24-channel input
rec = {"workload": "SSR_RRDBNet(nf=64,nb=23,gc=32) inference, 8xS2 (24-ch) 32x32 -> 128x128 chunks, bf16, inputs resident in HBM "
                   "(BASELINE.json configs[4] per-GPU share: 256 chunks per 16x16 tile)", "batch": B,
       "module_call": {"chunks_per_s": B / t_mod, "ms_per_batch": 1e3 * t_mod, "tiles_per_s": B / t_mod / 256},
       "launch_list": {"chunks_per_s": B / t_run, "ms_per_batch": 1e3 * t_run},
       "gflop_per_chunk": gflop, "tflops_launch_list": B / t_run * gflop / 1e3, "frac_of_bf16_mfma_peak": B / t_run * gflop / 1e3 / 2500.0}
print(json.dumps(rec))
print(f"generator inference, B={B}: module call {B / t_mod:.0f} img/s ({1e3 * t_mod:.2f} ms), launch list {B / t_run:.0f} img/s ({1e3 * t_run:.2f} ms)")
