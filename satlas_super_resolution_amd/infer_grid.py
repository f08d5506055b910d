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
    # PNG decode / encode off the critical path, in worker PROCESSES (png_io.py: threads do not scale, Pillow holds the GIL):
    # they read the NEXT batch's files while the device runs this one and compress + write the PREVIOUS batch's chunks.  Frame
    # selection (format_s2naip_data consumes the global `random` stream) stays in this process, in listing order, so the same
    # seed picks the same frames as a serial run.  `io_workers` (option file, our extension): processes per rank; 0 = threads.
    from . import png_io
    workers = int(opt.get("io_workers", max(1, min(16, (os.cpu_count() or 4) // max(1, world)))))
    done = 0

    def out_path(i):
        tile, idx = pngs[i].split("/")[-2], pngs[i].split("/")[-1]      # keep tile / index so that the stitch finds them
        return os.path.join(save_path, tile, idx)

    tiles = sorted(t for t in os.listdir(data_dir) if os.path.isdir(os.path.join(data_dir, t)))
    complete = {t for t in tiles if len(os.listdir(os.path.join(data_dir, t))) >= 256}
    keep = {t: {} for t in complete} if world == 1 else {}     # one rank: the mosaics are built from the arrays, not re-read from disk
    raw_keep = {t: {} for t in complete} if world == 1 else {}
    stitched = 0
    cells = [f"{i}_{j}.png" for i in range(16) for j in range(16)]

    def submit_stitch(pool, tile):
        """infer_grid.py:69-85 for one tile: stitched_sr.png (2048) and stitched_s2.png (512, first frame of every stack)"""
        sr_path, s2_path = os.path.join(save_path, tile, "stitched_sr.png"), os.path.join(save_path, tile, "stitched_s2.png")
        if tile in keep and len(keep[tile]) >= 256:
            fs = [pool.submit("stitch_and_save", [keep[tile][c] for c in cells], 2048, sr_path),
                  pool.submit("stitch_and_save", [raw_keep[tile][c] for c in cells], 512, s2_path, True)]
            keep.pop(tile), raw_keep.pop(tile)
            return fs
        return [pool.submit("stitch_from_dir", os.path.join(save_path, tile), 2048, sr_path),
                pool.submit("stitch_from_dir", os.path.join(data_dir, tile), 512, s2_path, True)]

    CH = 8                                        # files per task: amortises the hand-over to a worker
    chunked = lambda seq: [seq[k:k + CH] for k in range(0, len(seq), CH)]
    with png_io.PngWorkerPool(workers) as pool:
        groups = [mine[b0:b0 + batch] for b0 in range(0, len(mine), batch)]
        read_group = lambda idxs: [pool.submit("read_many", [pngs[i] for i in part]) for part in chunked(idxs)]
        reads = read_group(groups[0]) if groups else []
        saves, stitches = [], []
        stitched_now = set()                      # tiles stitched in THIS run (a stitched_sr.png on disk may be a stale one)
        for g, idxs in enumerate(groups):
            raw = [a for f in reads for a in f.result()]
            reads = read_group(groups[g + 1]) if g + 1 < len(groups) else []
            inputs = [format_s2naip_data(r, n_lr_images, "cpu")[0] for r in raw]
            out = infer_chunks(model, inputs, batch=len(inputs), device=device)
            saves += [pool.submit("save_many", [(out[k], out_path(idxs[k])) for k in part]) for part in chunked(list(range(len(idxs))))]
            done += len(idxs)
            for k, i in enumerate(idxs):          # a tile whose last chunk has just come off the device is stitched in the
                tile, idx = pngs[i].split("/")[-2], pngs[i].split("/")[-1]      # background while the next tiles run
                if tile in keep:
                    keep[tile][idx], raw_keep[tile][idx] = out[k], raw[k]
                    if len(keep[tile]) == 256 and all(c in keep[tile] for c in cells):
                        stitches += submit_stitch(pool, tile)
                        stitched_now.add(tile)
                        stitched += 1
        for f in saves + stitches:
            f.result()
        barrier()
        if rank == 0:
            rest = []
            for tile in tiles:
                if tile not in complete:
                    print("Tile ", tile, " contains less than 256 chunks, cannot stitch. Skipping.")
                    continue
                if tile in stitched_now:
                    continue
                rest += submit_stitch(pool, tile)      # several ranks wrote the chunks: read them back, tiles in parallel
                stitched += 1
            for f in rest:
                f.result()
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
