"""Whole-tile inference with the reference's command line and option file
(/root/reference/ssr/infer_grid.py:15-85, ssr/options/infer_grid_example.yml):

    python -m satlas_super_resolution_amd.infer_grid -opt infer_grid_example.yml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m satlas_super_resolution_amd.infer_grid -opt ...

Same inputs ({data_dir}/{tile}/{idx}.png, each [n_s2_images*32, 32, 3]), same outputs ({save_path}/{tile}/{idx}.png 128x128 chunks,
stitched_sr.png 2048x2048 and stitched_s2.png 512x512 per complete tile).  What changes is the execution: the reference pushes one
chunk at a time through the network; here the chunk list is sharded over the ranks (rank r takes chunks r, r + world, ...: no
collective on the data path), every rank runs its share in batches through the HIP generator, quantises on the
device (truncating uint8, infer_grid.py:60-64) and writes its own PNGs; rank 0 stitches after a barrier.
`compute_dtype` in the option file (our extension; default fp32h = the all-gates mode; its forward runs at fp32x3 speed) selects the arithmetic; `batch` the chunk batch."""
from __future__ import annotations

import argparse
import glob
import os
import random
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
    net_opt.setdefault("compute_dtype", opt.get("compute_dtype", "fp32h"))
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
    from .utils.infer_utils import frames_to_input, quantize_output, select_frames
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
    # worker budget: the cores this process may really use (affinity mask capped by the cgroup quota - os.cpu_count() says 128 on a
    # box that grants 16), shared by the ranks of the node, one core left to each rank's driver thread
    workers = int(opt.get("io_workers", max(1, min(16, png_io.host_cores() // max(1, world) - 1))))
    done = 0
    import time as _clock
    t_start = _clock.perf_counter()

    def out_path(i):
        tile, idx = pngs[i].split("/")[-2], pngs[i].split("/")[-1]      # keep tile / index so that the stitch finds them
        return os.path.join(save_path, tile, idx)

    tiles = sorted(t for t in os.listdir(data_dir) if os.path.isdir(os.path.join(data_dir, t)))
    complete = {t for t in tiles if len(os.listdir(os.path.join(data_dir, t))) >= 256}
    keep = {t: {} for t in complete} if world == 1 else {}     # one rank: the mosaics are built from the arrays, not re-read from disk
    raw_keep = {t: {} for t in complete} if world == 1 else {}
    stitched = 0
    cells = [f"{i}_{j}.png" for i in range(16) for j in range(16)]

    from collections import deque
    SLOT = 256 * 1024                             # bytes per decoded input stack in a block (85 frames; larger stacks come back as arrays)
    CH = max(4, -(-batch // max(1, min(workers, 8))))   # files per task: a batch is at most 8 tasks (each hand-over costs the driver thread a GIL round trip)
    chunked = lambda seq: [seq[k:k + CH] for k in range(0, len(seq), CH)]
    groups = [mine[b0:b0 + batch] for b0 in range(0, len(mine), batch)]
    on_gpu = torch.device(device).type == "cuda"
    out_bytes = batch * (4 * 32) * (4 * 32) * 3   # one batch of 128 x 128 chunks
    NIN, NOUT = 3, 6
    sdir = png_io.shm_dir(NIN * batch * SLOT + NOUT * out_bytes + 2048 * 2048 * 3 * 2 + 512 * 512 * 3 * 2)
    blocks = []

    def new_block(nbytes, tag):
        blocks.append(png_io.ShmBlock(nbytes, sdir, f"r{rank}_{tag}"))
        return blocks[-1]

    def submit_stitch(pool, tile):
        """infer_grid.py:69-85 for one tile: stitched_sr.png (2048) and stitched_s2.png (512, first frame of every stack).  One rank:
        the mosaics are assembled here from the arrays (two transposes) and handed to an encoder through a block of their own."""
        sr_path, s2_path = os.path.join(save_path, tile, "stitched_sr.png"), os.path.join(save_path, tile, "stitched_s2.png")
        if tile in keep and len(keep[tile]) >= 256:
            fs = []
            for arrs, size, path, s2 in ((keep.pop(tile), 2048, sr_path, False), (raw_keep.pop(tile), 512, s2_path, True)):
                m = png_io.mosaic([arrs[c] for c in cells], size, 16, s2)
                # a one-shot block per mosaic: not in `blocks` (it is closed and unlinked as soon as its file is written), and the
                # worker maps it transiently - a cached mapping of an unlinked block keeps its tmpfs pages allocated
                blk = png_io.ShmBlock(m.nbytes, sdir, f"r{rank}_mosaic")
                blk.buf[:] = m.reshape(-1)
                f = pool.submit("save_from", blk.path, blk.nbytes, [(0, tuple(m.shape), path)], True)
                f.add_done_callback(lambda _f, b=blk: b.close())
                fs.append(f)
            return fs
        return [pool.submit("stitch_from_dir", os.path.join(save_path, tile), 2048, sr_path),
                pool.submit("stitch_from_dir", os.path.join(data_dir, tile), 512, s2_path, True)]

    import contextlib
    with contextlib.ExitStack() as _stack:
        _stack.callback(lambda: [b.close() for b in blocks])          # registered first: runs after the pool has shut down
        pool = _stack.enter_context(png_io.shared_pool(workers))      # the workers stay up for the next call of this process
        in_blocks = [new_block(batch * SLOT, f"in{k}") for k in range(NIN)]
        out_blocks = [new_block(out_bytes, f"out{k}") for k in range(NOUT)]
        out_busy = [[] for _ in range(NOUT)]      # save tasks that still read the block

        def read_group(g):
            """decode tasks of batch g into input block g % NIN (free again: batch g - NIN has been picked apart)"""
            blk = in_blocks[g % NIN]
            return blk, [pool.submit("read_into", [pngs[i] for i in part], blk.path, blk.nbytes, [SLOT * (k0 + k) for k in range(len(part))], SLOT)
                         for k0, part in ((k0, groups[g][k0:k0 + CH]) for k0 in range(0, len(groups[g]), CH))]

        pinned = [torch.empty(batch, n_lr_images, 32, 32, 3, dtype=torch.uint8, pin_memory=on_gpu) for _ in range(2)]
        prep_n = [0]

        def prep(rd):
            """decoded stacks of a batch -> (first frames, uint8 [B, n, 32, 32, 3] of the chosen frames in a pinned buffer).  Which
            frames: select_frames' rule (format_s2naip_data), `random` consumed chunk by chunk in listing order; the zero test
            runs once over the whole batch when the stacks have one shape (they do in a Sentinel-2 tile)."""
            blk, futs = rd
            shapes = [shp for f in futs for shp in f.result()]
            B = len(shapes)
            dst = pinned[prep_n[0] & 1][:B]
            prep_n[0] += 1
            dst_np = dst.numpy()
            if all(isinstance(sh, tuple) and sh == shapes[0] for sh in shapes) and int(np.prod(shapes[0])) % 3072 == 0:
                T = int(np.prod(shapes[0])) // 3072
                stacks = np.lib.stride_tricks.as_strided(blk.buf, shape=(B, T, 32, 32, 3), strides=(SLOT, 3072, 96, 3, 1), writeable=False)
                has_zero = (stacks == 0).any(axis=(2, 3, 4))
                for k in range(B):
                    clean = np.flatnonzero(~has_zero[k]).tolist()
                    if len(clean) >= n_lr_images:
                        chosen = random.sample(clean, n_lr_images)
                    else:
                        chosen = clean + random.sample(np.flatnonzero(has_zero[k]).tolist(), n_lr_images - len(clean))
                    dst_np[k] = stacks[k, chosen]
                firsts = [np.array(stacks[k, 0]) for k in range(B)]
            else:
                firsts = []
                for k, shp in enumerate(shapes):
                    a = shp if isinstance(shp, np.ndarray) else blk.buf[SLOT * k:SLOT * k + int(np.prod(shp))].reshape(shp)
                    dst_np[k], first = select_frames(a, n_lr_images)
                    firsts.append(np.array(first))
            return firsts, dst

        def launch(sel):
            """enqueue upload, formatting, generator, truncating uint8 and the download of one batch; nothing waits here"""
            with torch.no_grad():
                y = model(frames_to_input(sel.to(device, non_blocking=True)))
                if not on_gpu:
                    return quantize_output(y), None
                from .metrics import tensor2img_u8
                yq = tensor2img_u8(y.detach(), truncate=True)
                yh = torch.empty(yq.shape, dtype=torch.uint8, pin_memory=True)
                yh.copy_(yq, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                return yh, ev

        # software pipeline over the batches: the files of batches g + 1 and g + 2 are being decoded while the device runs batch g and
        # this thread picks the frames of batch g + 1; batch g's chunks go to the encoders (through an output block) as soon as its
        # download event has fired
        saves, stitches = [], []
        stitched_now = set()                      # tiles stitched in THIS run (a stitched_sr.png on disk may be a stale one)
        rq = deque(read_group(g) for g in range(min(2, len(groups))))
        nxt = prep(rq.popleft()) if groups else None
        import time as _time
        trace = [] if os.environ.get("SSR_INFER_TRACE") == "1" else None
        for g, idxs in enumerate(groups):
            firsts, sel = nxt
            t0 = _time.perf_counter()
            pending = launch(sel)
            t1 = _time.perf_counter()
            if g + 2 < len(groups):
                rq.append(read_group(g + 2))
            if g + 1 < len(groups):
                nxt = prep(rq.popleft())
            t2 = _time.perf_counter()
            yh, ev = pending
            if ev is not None:
                ev.synchronize()
                yh = yh.numpy()
            t3 = _time.perf_counter()
            ob = g % NOUT
            for f in out_busy[ob]:                # the block's previous batch has been written out
                f.result()
            if trace is not None:
                trace.append((g, t1 - t0, t2 - t1, t3 - t2, _time.perf_counter() - t3))
            oblk = out_blocks[ob]
            n_out = yh[0].nbytes
            oblk.buf[:len(idxs) * n_out] = yh.reshape(-1)
            out_busy[ob] = [pool.submit("save_from", oblk.path, oblk.nbytes,
                                        [(k * n_out, tuple(yh[k].shape), out_path(idxs[k])) for k in part])
                            for part in chunked(list(range(len(idxs))))]
            saves += out_busy[ob]
            done += len(idxs)
            for k, i in enumerate(idxs):          # a tile whose last chunk has just come off the device is stitched in the
                tile, idx = pngs[i].split("/")[-2], pngs[i].split("/")[-1]      # background while the next tiles run
                if tile in keep:
                    keep[tile][idx], raw_keep[tile][idx] = np.array(yh[k]), firsts[k]
                    if len(keep[tile]) == 256 and all(c in keep[tile] for c in cells):
                        stitches += submit_stitch(pool, tile)
                        stitched_now.add(tile)
                        stitched += 1
        if trace:
            for g, a, b, c, d in trace[:24]:
                print(f"[infer trace] batch {g}: launch {1e3 * a:.1f} ms, reads+prep of next {1e3 * b:.1f}, wait device {1e3 * c:.1f}, wait out block {1e3 * d:.1f}")
        _t = _time.perf_counter()
        for f in saves:
            f.result()
        _t1 = _time.perf_counter()
        for f in stitches:
            f.result()
        if trace is not None:
            print(f"[infer trace] after the loop: chunk files {1e3 * (_t1 - _t):.1f} ms, mosaics {1e3 * (_time.perf_counter() - _t1):.1f} ms")
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
    secs = _clock.perf_counter() - t_start
    # (io_workers / seconds: what the run had and took - eight ranks on one host share its cores, README.md:159 of the reference)
    return {"chunks": done, "tiles_stitched": stitched, "io_workers": workers, "seconds": round(secs, 3)}


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
