"""Inference-side host helpers with the reference's names and semantics
(/root/reference/ssr/utils/infer_utils.py:6-60, /root/reference/ssr/infer_grid.py:46-85), plus the batched, rank-sharded
chunk loop the reference runs one 32x32 chunk at a time.

The chunk loop shards naturally: every chunk is independent, so rank r takes chunks r, r+world, ... with
no data-path collective; results are gathered on the host for `stitch`.
"""
import os
import random
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch


def format_s2naip_data(s2_data, n_s2_images: int, device):
    """infer_utils.py:6-39.  s2_data: uint8 [T*32, 32, 3] (T stacked Sentinel-2 frames of one 32x32 chunk).  Picks
    n_s2_images frames with `random.sample`, preferring frames without any black ([0,0,0]) pixel, and returns
    (float tensor [1, n_s2_images*3, 32, 32] in [0,1] on `device`, the first frame as uint8 [32,32,3])."""
    s2_chunks = np.reshape(s2_data, (-1, 32, 32, 3))
    s2_image = s2_chunks[0]
    goods, bads = [], []
    for i, ts in enumerate(s2_chunks):
        # `[0, 0, 0] in ts` on an ndarray is (ts == [0,0,0]).any(): true as soon as ANY channel value is 0 (:17)
        if (ts == np.array([0, 0, 0])).any():
            bads.append(i)
        else:
            goods.append(i)
    if len(goods) >= n_s2_images:
        rand_indices = random.sample(goods, n_s2_images)
    else:
        need = n_s2_images - len(goods)
        rand_indices = goods + random.sample(bads, need)
    picked = np.array([s2_chunks[i] for i in rand_indices])
    chunks = [torch.as_tensor(img).permute(2, 0, 1) for img in picked]
    s2_tensor = torch.cat(chunks).unsqueeze(0)
    s2_tensor = s2_tensor.to(device).float() / 255
    return s2_tensor, s2_image


def quantize_output(output: torch.Tensor) -> np.ndarray:
    """infer_grid.py:60-64 / infer.py: clamp(0,1) -> *255 -> astype(uint8) (truncation), NCHW -> [N,H,W,3]."""
    out = torch.clamp(output, 0, 1).detach().float().cpu().numpy()
    return np.transpose(out * 255, (0, 2, 3, 1)).astype(np.uint8)


def stitch_arrays(chunks: Dict, img_size: int, grid_size: int = 16, sentinel2: bool = False) -> np.ndarray:
    """The paste loop of infer_utils.stitch (:41-60) on in-memory chunks: chunks[(i, j)] is the uint8 image of grid cell
    row i, column j ([n*32,32,3] Sentinel-2 stacks contribute their first frame when sentinel2=True)."""
    chunk_size = int(img_size / grid_size)
    empty = np.zeros((img_size, img_size, 3))
    for i in range(grid_size):
        for j in range(grid_size):
            load = np.asarray(chunks[(i, j)])
            if sentinel2:
                load = np.reshape(load, (-1, 32, 32, 3))[0]
            empty[i * chunk_size:i * chunk_size + chunk_size, j * chunk_size:j * chunk_size + chunk_size, :] = load
    return empty.astype(np.uint8)


def stitch(chunks_dir: str, img_size: int, save_path: str, scale: int = 4, grid_size: int = 16, sentinel2: bool = False):
    """infer_utils.py:41-60 with the same signature: reads `{chunks_dir}/{i}_{j}.png`, writes the stitched PNG.  PNG I/O
    through Pillow (the reference uses skimage.io, which is not a dependency of this package)."""
    from PIL import Image
    chunks = {(i, j): np.asarray(Image.open(os.path.join(chunks_dir, f"{i}_{j}.png")).convert("RGB"))
              for i in range(grid_size) for j in range(grid_size)}
    Image.fromarray(stitch_arrays(chunks, img_size, grid_size, sentinel2)).save(save_path)


@torch.no_grad()
def infer_chunks(model: Callable[[torch.Tensor], torch.Tensor], inputs: Sequence[torch.Tensor], batch: int = 64,
                 rank: int = 0, world: int = 1, device: Optional[torch.device] = None) -> Dict[int, np.ndarray]:
    """The model loop of infer_grid.py:46-64 over a list of formatted chunk tensors ([1, C, 32, 32] each), batched and
    sharded: this rank processes chunks rank, rank+world, ... in batches of `batch` and returns {chunk index: uint8
    [4h, 4w, 3]} for its share.  No collective: the caller gathers dicts on the host (torch.distributed.gather_object or
    files) before `stitch`."""
    mine: List[int] = list(range(rank, len(inputs), world))
    out: Dict[int, np.ndarray] = {}
    for b0 in range(0, len(mine), batch):
        idx = mine[b0:b0 + batch]
        x = torch.cat([inputs[i] for i in idx], 0)
        if device is not None:
            x = x.to(device)
        y = quantize_output(model(x))
        for k, i in enumerate(idx):
            out[i] = y[k]
    return out
