"""Whole-tile inference with the reference's command line and option file
(/root/reference/ssr/infer_grid.py:15-85, ssr/options/infer_grid_example.yml):

    python -m satlas_super_resolution_amd.infer_grid -opt infer_grid_example.yml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m satlas_super_resolution_amd.infer_grid -opt ...

Same inputs ({data_dir}/{tile}/{idx}.png, each [n_s2_images*32, 32, 3]), same outputs ({save_path}/{tile}/{idx}.png 128x128 chunks,
stitched_sr.png 2048x2048 and stitched_s2.png 512x512 per complete tile).  What changes is the execution: the reference pushes one
chunk at a time through the network; here the chunk list is sharded over the ranks (rank r takes chunks r, r + world, ...: no
collective on the data path), every rank runs its share in batches through the HIP generator, quantises on the
device (truncating uint8, infer_grid.py:60-64) and writes its own PNGs; rank 0 stitches after a barrier.
`compute_dtype` in the option file (our extension; default fp32x3 = the parity mode) selects the arithmetic; `batch` the chunk batch."""
from __future__ import annotations

import argparse
import glob
import os
from typing import Callable, Dict, Optional

import numpy as np
import torch


def _read_png(path: str) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"))


def load_generator(opt: Dict, device) -> torch.nn.Module:
    """infer_grid.py:30-40: build_network(opt) + the weights named by the option file."""
    from .utils.model_utils import build_network
    opt = dict(opt)
    net_opt = dict(opt["network_g"])
    net_opt.setdefault("compute_dtype", opt.get("compute_dtype", "fp32x3"))
    opt["network_g"] = net_opt
    model = build_network(opt)
    path = opt.get("path", {})
    if "pretrain_network_g" not in path:
        print("WARNING: Model weights are not specified in configuration file.")
    else:
        state_dict = torch.load(path["pretrain_network_g"], map_location="cpu")
        model.load_state_dict(state_dict[path["param_key_g"]], strict=path["strict_load_g"])
    model = model.to(device).eval()
    if torch.device(device).type == "cuda":
        model.freeze_packed()          # fixed weights for the whole run: pack once, not per chunk batch (HipNet.pack_if_stale)
    return model


def run_infer_grid(opt: Dict, model: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, rank: int = 0, world: int = 1,
                   device=None, barrier: Callable[[], None] = lambda: None) -> Dict[str, int]:
    from .utils.infer_utils import format_s2naip_data, infer_chunks, stitch
    data_dir, n_lr_images, save_path = opt["data_dir"], opt["n_lr_images"], opt["save_path"]
    if device is None:
        device = torch.device("cuda")
    if model is None:
        model = load_generator(opt, device)
    pngs = sorted(glob.glob(data_dir + "/**/*.png", recursive=True))      # sorted: every rank must see the same order
    if rank == 0:
        print("Running inference on ", len(pngs), " images.")
    mine = list(range(rank, len(pngs), world))
    batch = int(opt.get("batch", 64))
    done = 0
    for b0 in range(0, len(mine), batch):
        idxs = mine[b0:b0 + batch]
        inputs = [format_s2naip_data(_read_png(pngs[i]), n_lr_images, "cpu")[0] for i in idxs]
        out = infer_chunks(model, inputs, batch=len(inputs), device=device)
        from PIL import Image
        for k, i in enumerate(idxs):
            tile, idx = pngs[i].split("/")[-2], pngs[i].split("/")[-1]      # keep tile / index so that the stitch finds them
            os.makedirs(os.path.join(save_path, tile), exist_ok=True)
            Image.fromarray(out[k]).save(os.path.join(save_path, tile, idx))
            done += 1
    barrier()
    stitched = 0
    if rank == 0:
        for tile in sorted(os.listdir(data_dir)):
            if not os.path.isdir(os.path.join(data_dir, tile)):
                continue
            if len(os.listdir(os.path.join(data_dir, tile))) < 256:
                print("Tile ", tile, " contains less than 256 chunks, cannot stitch. Skipping.")
                continue
            stitch(os.path.join(save_path, tile), 2048, os.path.join(save_path, tile, "stitched_sr.png"))
            stitch(os.path.join(data_dir, tile), 512, os.path.join(save_path, tile, "stitched_s2.png"), sentinel2=True)
            stitched += 1
    return {"chunks": done, "tiles_stitched": stitched}


def main():
    import yaml
    parser = argparse.ArgumentParser()
    parser.add_argument("-opt", type=str, help="Path to the options file.")
    args = parser.parse_args()
    with open(args.opt) as f:
        opt = yaml.safe_load(f)
    from .dp import init_distributed
    ctx = init_distributed()
    res = run_infer_grid(opt, rank=ctx.rank, world=ctx.world, barrier=ctx.barrier)
    print(f"rank {ctx.rank}: {res}")


if __name__ == "__main__":
    main()
