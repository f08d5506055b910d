"""Per-image inference with the reference's command line and option file (/root/reference/ssr/infer.py:14-67,
ssr/options/infer_example.yml):

    python -m satlas_super_resolution_amd.infer -opt infer_example.yml

Same inputs (every PNG under `data_dir`, each [n_s2_images*32, 32, 3]) and the same outputs: `{save_path}/{i}/lr.png` (the first
Sentinel-2 frame, for comparison) and `{save_path}/{i}/sr.png` (clamp(0,1) * 255 truncated to uint8, infer.py:58-61), i = position of
the image in the listing.  The listing is sorted here (the reference takes glob's order, which is the file system's); the images
go through the HIP generator in batches instead of one at a time; `compute_dtype` (default fp32h, the all-gates mode) and `batch` are
this package's two extra option keys."""
from __future__ import annotations

import argparse
import glob
import os
from typing import Callable, Dict, Optional

import torch


def run_infer(opt: Dict, model: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, device=None) -> Dict[str, int]:
    from PIL import Image
    from .infer_grid import _read_png, load_generator
    from .utils.infer_utils import format_s2naip_data, infer_chunks
    data_dir, n_lr_images, save_path = opt["data_dir"], opt["n_lr_images"], opt["save_path"]
    if device is None:
        device = torch.device("cuda")
    if model is None:
        model = load_generator(opt, device)
    pngs = sorted(glob.glob(data_dir + "/**/*.png", recursive=True))
    print("Running inference on ", len(pngs), " images.")
    batch = int(opt.get("batch", 64))
    for b0 in range(0, len(pngs), batch):
        idxs = list(range(b0, min(b0 + batch, len(pngs))))
        formatted = [format_s2naip_data(_read_png(pngs[i]), n_lr_images, "cpu") for i in idxs]
        out = infer_chunks(model, [f[0] for f in formatted], batch=len(idxs), device=device)
        for k, i in enumerate(idxs):
            save_dir = os.path.join(save_path, str(i))
            os.makedirs(save_dir, exist_ok=True)
            Image.fromarray(formatted[k][1]).save(save_dir + "/lr.png")       # the low-res input beside the result (infer.py:55-57)
            Image.fromarray(out[k]).save(save_dir + "/sr.png")
    return {"images": len(pngs)}


def main():
    import yaml
    parser = argparse.ArgumentParser()
    parser.add_argument("-opt", type=str, help="Path to the options file.")
    args = parser.parse_args()
    with open(args.opt) as f:
        opt = yaml.safe_load(f)
    print(run_infer(opt))


if __name__ == "__main__":
    main()
