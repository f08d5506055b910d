#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/tr.py <<'PY'
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from PIL import Image
from satlas_super_resolution_amd.archs.rrdbnet_arch import SSR_RRDBNet
from satlas_super_resolution_amd.infer_grid import run_infer_grid
from satlas_super_resolution_amd import png_io
tmp = tempfile.mkdtemp(prefix="tr_")
rng = np.random.RandomState(0); yy, xx = np.mgrid[0:256, 0:32]
paths = []
for t in range(3):
    d = os.path.join(tmp, "in", f"tile{t}"); os.makedirs(d)
    for i in range(16):
        for j in range(16):
            img = 110 + 60 * np.sin((yy + 7 * i) / 19.0)[..., None] * np.cos((xx + 5 * j) / 11.0)[..., None] + rng.randint(-12, 13, (256, 32, 3))
            p = os.path.join(d, f"{i}_{j}.png"); Image.fromarray(np.clip(img, 1, 255).astype(np.uint8)).save(p); paths.append(p)
# pool micro-benchmark: latency of one batch of read tasks on an idle pool
blk = png_io.ShmBlock(64 * 262144, png_io.shm_dir(1 << 26), "probe")
for w in (15, 8, 4):
    with png_io.PngWorkerPool(w) as pool:
        for rep in range(3):
            t0 = time.perf_counter()
            fs = [pool.submit("read_into", paths[k:k + 8], blk.path, blk.nbytes, [262144 * (k + q) for q in range(8)], 262144) for k in range(0, 64, 8)]
            [f.result() for f in fs]
            t1 = time.perf_counter() - t0
        outs = np.zeros((64, 128, 128, 3), np.uint8)
        print(f"pool {w}: one batch of 8 read tasks {1e3 * t1:.1f} ms")
blk.close()
net = SSR_RRDBNet(24, 3, 4, 64, 23, 32, compute_dtype="fp32x3").cuda().eval().freeze_packed()
opt = {"data_dir": os.path.join(tmp, "in") + "/", "n_lr_images": 8, "save_path": os.path.join(tmp, "out") + "/", "batch": 64, "io_workers": int(sys.argv[1])}
run_infer_grid(dict(opt, save_path=os.path.join(tmp, "warm") + "/"), model=net)
os.environ["SSR_INFER_TRACE"] = "1"
t0 = time.perf_counter(); run_infer_grid(opt, model=net); print("total", time.perf_counter() - t0)
# how long do the encoders take on this box?
import io
a = np.clip(110 + 60 * np.sin(np.mgrid[0:128, 0:128][0] / 19.0)[..., None] + rng.randint(-12, 13, (128, 128, 3)), 0, 255).astype(np.uint8)
t0 = time.perf_counter()
for _ in range(100): b = png_io.encode_png(a)
print("encode chunk", (time.perf_counter() - t0) * 10, "ms")
os.makedirs(os.path.join(tmp, "w"), exist_ok=True)
t0 = time.perf_counter()
for k in range(200): png_io.save_png(a, os.path.join(tmp, "w", f"{k}.png"))
print("save_png chunk", (time.perf_counter() - t0) * 5, "ms")
big = np.tile(a, (16, 16, 1)); t0 = time.perf_counter(); png_io.save_png(big, os.path.join(tmp, "w", "m.png")); print("save mosaic", (time.perf_counter() - t0) * 1e3, "ms")
with png_io.PngWorkerPool(15) as pool:
    blk = png_io.ShmBlock(64 * 49152, png_io.shm_dir(1 << 26), "probe2")
    for rep in range(3):
        t0 = time.perf_counter()
        fs = [pool.submit("timed", "save_from", blk.path, blk.nbytes, [(49152 * (k + q), (128, 128, 3), os.path.join(tmp, "w2", f"{rep}_{k+q}.png")) for q in range(8)]) for k in range(0, 64, 8)]
        rs = [f.result() for f in fs]
        t1 = time.perf_counter() - t0
    print(f"pool 15: 8 save tasks of 8 chunks {1e3 * t1:.1f} ms; per task on the worker {[round(1e3 * (r[1] - r[0]), 1) for r in rs]}")
    blk.close()
shutil.rmtree(tmp, ignore_errors=True)
PY
python /tmp/tr.py 15 2>/dev/null | grep -v Running > $O/r04k_trace.txt
cat $O/r04k_trace.txt
