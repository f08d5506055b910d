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
    n_s2_images frames with `random.sample`, preferring frames without a zero-valued sample, and returns
    (float tensor [1, n_s2_images*3, 32, 32] in [0,1] on `device`, the first frame as uint8 [32,32,3]).

    The reference's validity test `[0, 0, 0] in frame` on an ndarray is `(frame == [0, 0, 0]).any()`: a frame is set aside
    as soon as ANY of its values is 0 (:17).  The `random` module is consumed exactly as the reference does (one
    `random.sample` over the index list of the same length), so the same seed picks the same frames."""
    chosen, first = select_frames(s2_data, n_s2_images)
    stack = torch.from_numpy(chosen)                                                # [n, 32, 32, 3] uint8
    s2_tensor = stack.permute(0, 3, 1, 2).reshape(1, n_s2_images * 3, 32, 32)      # frame-major, RGB inside a frame
    return s2_tensor.to(device).float() / 255, first


def select_frames(s2_data, n_s2_images: int):
    """The host half of format_s2naip_data (infer_utils.py:6-32): which frames of the stack go to the network - the `random`
    module consumed exactly as the reference consumes it.  Returns (uint8 [n_s2_images, 32, 32, 3], the first frame)."""
    frames = np.reshape(s2_data, (-1, 32, 32, 3))
    has_zero = (frames == 0).any(axis=(1, 2, 3))
    clean, dirty = np.flatnonzero(~has_zero).tolist(), np.flatnonzero(has_zero).tolist()
    if len(clean) >= n_s2_images:
        chosen = random.sample(clean, n_s2_images)
    else:
        chosen = clean + random.sample(dirty, n_s2_images - len(clean))
    return np.ascontiguousarray(frames[chosen]), frames[0]


def frames_to_input(sel_u8: torch.Tensor) -> torch.Tensor:
    """The device half, for a whole batch at once: uint8 [B, n, 32, 32, 3] -> float [B, n*3, 32, 32] in [0, 1]; per chunk the
    same permute / reshape / `.float() / 255` as format_s2naip_data (:33-38), so the values are bit-identical."""
    b, n = sel_u8.shape[:2]
    return sel_u8.permute(0, 1, 4, 2, 3).reshape(b, n * 3, 32, 32).float() / 255


def quantize_output(output: torch.Tensor) -> np.ndarray:
    """infer_grid.py:60-64 / infer.py:58-60: clamp(0,1) -> *255 -> astype(uint8) (truncation), NCHW -> uint8 [N,H,W,C] on
    the host.  Device tensors are quantised on the device (ssr_quantize_u8, csrc/metrics.hip) and only bytes cross PCIe."""
    if output.is_cuda:
        from ..metrics import tensor2img_u8
        return tensor2img_u8(output.detach(), truncate=True).cpu().numpy()
    out = torch.clamp(output, 0, 1).detach().float().numpy()
    return np.transpose(out * 255, (0, 2, 3, 1)).astype(np.uint8)


def stitch_arrays(chunks: Dict, img_size: int, grid_size: int = 16, sentinel2: bool = False) -> np.ndarray:
    """infer_utils.stitch (:41-60) on in-memory chunks: cell (i, j) lands at rows i*cs.., columns j*cs.. with
    cs = int(img_size / grid_size); [n*32,32,3] Sentinel-2 stacks contribute their first frame when sentinel2=True.
    One gather + one transpose instead of 256 slice assignments."""
    cs = int(img_size / grid_size)
    first = (lambda a: np.reshape(a, (-1, 32, 32, 3))[0]) if sentinel2 else (lambda a: a)
    tiles = np.stack([first(np.asarray(chunks[(i, j)])) for i in range(grid_size) for j in range(grid_size)])
    assert tiles.shape[1:] == (cs, cs, 3), (tiles.shape, cs)
    mosaic = tiles.reshape(grid_size, grid_size, cs, cs, 3).transpose(0, 2, 1, 3, 4).reshape(grid_size * cs, grid_size * cs, 3)
    canvas = np.zeros((img_size, img_size, 3), np.uint8)      # img_size not divisible by grid_size: zero margin, as the reference
    canvas[:grid_size * cs, :grid_size * cs] = mosaic
    return canvas


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
