"""BASELINE.json configs[4] on ONE GPU, at the mode it ships in: a whole 16 x 16 Sentinel-2 tile grid (256 chunks of 8 frames,
PNG files on disk) -> satlas_super_resolution_amd.infer_grid.run_infer_grid (PNG decode, frame selection, HIP generator in
batches, truncating uint8 on the device, PNG encode of the 256 chunks, both stitched mosaics) — wall time of the whole tile,
beside the generator alone (inputs resident in HBM) in the same arithmetic mode.

    python tools/infer_e2e_bench.py [fp32x3|bf16|fp32] [tiles] > profiles/r03_infer_e2e_<mode>.json
"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fp32x3"
    n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from PIL import Image
    from satlas_super_resolution_amd.archs.rrdbnet_arch import SSR_RRDBNet
    from satlas_super_resolution_amd.infer_grid import run_infer_grid
    from satlas_super_resolution_amd import png_io
    tmp = tempfile.mkdtemp(prefix="infer_e2e_")
    try:
        rng = np.random.RandomState(0)
        yy, xx = np.mgrid[0:256, 0:32]
        for t in range(n_tiles):
            d = os.path.join(tmp, "in", f"tile{t}")
            os.makedirs(d)
            for i in range(16):
                for j in range(16):      # smooth field + noise: PNG sizes like real imagery rather than incompressible noise
                    img = 110 + 60 * np.sin((yy + 7 * i) / 19.0)[..., None] * np.cos((xx + 5 * j) / 11.0)[..., None] + rng.randint(-12, 13, (256, 32, 3))
                    Image.fromarray(np.clip(img, 1, 255).astype(np.uint8)).save(os.path.join(d, f"{i}_{j}.png"))
        net = SSR_RRDBNet(24, 3, 4, 64, 23, 32, compute_dtype=mode).cuda().eval().freeze_packed()
        opt = {"data_dir": os.path.join(tmp, "in") + "/", "n_lr_images": 8, "save_path": os.path.join(tmp, "out") + "/", "batch": 64}
        n_workers = max(1, min(16, png_io.host_cores() - 1))
        t0 = time.perf_counter()
        with png_io.shared_pool(n_workers) as pool:      # the workers stay up for every later run_infer_grid call of this process
            [f.result() for f in [pool.submit("read_many", []) for _ in range(2 * n_workers)]]
        t_pool = time.perf_counter() - t0
        run_infer_grid(dict(opt, save_path=os.path.join(tmp, "warm") + "/"), model=net)      # warm-up: plans, first touch, page cache
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run_infer_grid(opt, model=net)
        torch.cuda.synchronize()
        t_e2e = time.perf_counter() - t0
        assert (res["chunks"], res["tiles_stitched"]) == (256 * n_tiles, n_tiles), res
        # the same run with the PNG work on threads of this process (io_workers = 0: the first version of the driver)
        shutil.rmtree(os.path.join(tmp, "out"), ignore_errors=True)
        t0 = time.perf_counter()
        run_infer_grid(dict(opt, io_workers=0), model=net)
        torch.cuda.synchronize()
        t_thr = time.perf_counter() - t0
        x = torch.rand(64, 24, 32, 32, device="cuda")
        with torch.no_grad():
            for _ in range(2):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                net(x)
            torch.cuda.synchronize()
            t_model = (time.perf_counter() - t0) / 8
        serial = None
        rec = {"workload": "BASELINE.json configs[4], one GPU's share: 16x16 grid of 8xS2 chunks (PNG on local disk) -> per-chunk 128x128 PNGs + "
                           "stitched_sr.png 2048x2048 + stitched_s2.png 512x512; SSR_RRDBNet(nf=64,nb=23,gc=32), random weights",
               "compute_dtype": mode, "tiles": n_tiles, "batch": 64, "io_workers": max(1, min(16, png_io.host_cores() - 1)), "io": "worker processes fed through shared memory (satlas_super_resolution_amd/png_io.py: own PNG writer, filter none + zlib 1); decode / frame selection / encode pipelined with the device; generator forward replayed as a hipGraph; the worker pool is started once per process (pool_startup_s, not in the per-tile time)", "pool_startup_s": t_pool,
               "end_to_end": {"seconds_per_tile": t_e2e / n_tiles, "tiles_per_s": n_tiles / t_e2e, "chunks_per_s": 256 * n_tiles / t_e2e},
               "end_to_end_threads": {"seconds_per_tile": t_thr / n_tiles, "tiles_per_s": n_tiles / t_thr},
               "generator_only": {"ms_per_64_chunks": 1e3 * t_model, "chunks_per_s": 64 / t_model, "tiles_per_s": 64 / t_model / 256,
                                  "tflops": 64 / t_model * 36.739 / 1e3},
               "host_cores": png_io.host_cores()}
        print(json.dumps(rec))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
