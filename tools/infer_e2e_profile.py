"""cProfile of the driver thread of run_infer_grid on the GPU box (tools/infer_e2e_bench.py's workload): where the host time goes."""
import cProfile, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from satlas_super_resolution_amd.archs.rrdbnet_arch import SSR_RRDBNet
from satlas_super_resolution_amd.infer_grid import run_infer_grid
mode, n_tiles, workers = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tmp = tempfile.mkdtemp(prefix="infer_prof_")
rng = np.random.RandomState(0)
yy, xx = np.mgrid[0:256, 0:32]
for t in range(n_tiles):
    d = os.path.join(tmp, "in", f"tile{t}")
    os.makedirs(d)
    for i in range(16):
        for j in range(16):
            img = 110 + 60 * np.sin((yy + 7 * i) / 19.0)[..., None] * np.cos((xx + 5 * j) / 11.0)[..., None] + rng.randint(-12, 13, (256, 32, 3))
            Image.fromarray(np.clip(img, 1, 255).astype(np.uint8)).save(os.path.join(d, f"{i}_{j}.png"))
net = SSR_RRDBNet(24, 3, 4, 64, 23, 32, compute_dtype=mode).cuda().eval().freeze_packed()
opt = {"data_dir": os.path.join(tmp, "in") + "/", "n_lr_images": 8, "save_path": os.path.join(tmp, "out") + "/", "batch": 64, "io_workers": workers}
run_infer_grid(dict(opt, save_path=os.path.join(tmp, "warm") + "/"), model=net)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
run_infer_grid(opt, model=net)
pr.disable()
dt = time.perf_counter() - t0
print(f"== {mode} workers={workers}: {dt / n_tiles:.3f} s per tile")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
shutil.rmtree(tmp, ignore_errors=True)
