"""Input-pipeline throughput: S2NAIPDataset (Pillow PNG decode + the reference's frame selection) behind torch's DataLoader, on a
synthetic dataset of the published shapes (NAIP 128x128x3, Sentinel-2 tci [T*32, 32, 3], T frames).  Prints one JSON line:
samples/s for 0..W workers on THIS host's cores.  The train step consumes ~2.3k img/s per GPU (bench.py), i.e. 8 workers per GPU
(`num_worker_per_gpu: 8`, esrgan_s2naip_urban.yml:29) must deliver ~290 samples/s each.
    python tools/loader_bench.py [--chips 256] [--frames 16] [--workers 0 4 8]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_dataset(root, chips, frames, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    base = rng.randint(1, 256, (512, 512, 3)).astype(np.uint8)
    for c in range(chips):
        chip = f"{1000 + c}_{2000 + c}"
        os.makedirs(os.path.join(root, "naip", chip), exist_ok=True)
        os.makedirs(os.path.join(root, "sentinel2", chip), exist_ok=True)
        y, x = rng.randint(0, 384, 2)
        Image.fromarray(base[y:y + 128, x:x + 128]).save(os.path.join(root, "naip", chip, chip + ".png"))     # noise: worst case for PNG
        Image.fromarray(rng.randint(1, 256, (frames * 32, 32, 3)).astype(np.uint8)).save(os.path.join(root, "sentinel2", chip, "tci.png"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chips", type=int, default=256)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--workers", type=int, nargs="*", default=[0, 4, 8])
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    import torch
    from satlas_super_resolution_amd.data import S2NAIPDataset
    from satlas_super_resolution_amd.data.s2naip_dataset import build_train_loader
    out = {"chips": args.chips, "frames": args.frames, "batch": args.batch, "host_cores": len(os.sched_getaffinity(0)), "samples_per_s": {}}
    with tempfile.TemporaryDirectory() as root:
        make_dataset(root, args.chips, args.frames)
        opt = {"phase": "train", "scale": 4, "name": "synthetic", "n_s2_images": 8, "sentinel2_path": os.path.join(root, "sentinel2"),
               "naip_path": os.path.join(root, "naip"), "batch_size_per_gpu": args.batch, "use_shuffle": True}
        ds = S2NAIPDataset(opt)
        for w in args.workers:
            opt["num_worker_per_gpu"] = w
            loader = build_train_loader(ds, opt)
            n, t0 = 0, None
            for ep in range(args.epochs + 1):
                for b in loader:
                    if ep > 0:
                        n += b["lr"].shape[0]
                if ep == 0:
                    t0 = time.perf_counter()          # first epoch: worker start-up and page cache
            out["samples_per_s"][str(w)] = round(n / (time.perf_counter() - t0), 1)
            assert b["lr"].dtype == torch.uint8 and tuple(b["lr"].shape[1:]) == (24, 32, 32) and tuple(b["hr"].shape[1:]) == (3, 128, 128)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
