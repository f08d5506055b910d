"""S2NAIPDataset — the training / validation dataset of the ESRGAN option files, with the reference's registry name, `opt` keys
and sample dictionary (/root/reference/ssr/data/s2-naip_dataset.py:35-246):

    {'hr': uint8 [3,128,128], 'lr': uint8 [n_s2_images*C, 32, 32] (or [n, C, 32, 32] with use_3d), ['old_hr': uint8 [3,128,128]],
     'Index': int, 'Phase': str, 'Chip': str}

uint8 all the way: SSRESRGANModel.feed_data uploads bytes and scales by 1/255 on the device (0.8 MB per batch of 32 instead of
3.2 MB of fp32), so the loader's cost is PNG decode + frame selection.  PNG files are decoded with Pillow (the reference uses
torchvision.io.read_image, which is not a dependency here); the frame selection consumes Python's `random` exactly as the reference
does (one `random.sample`, then one `random.randint` under rand_crop), so a seeded run picks the same frames.

Not implemented (raise): OSM-object filtering (`osm_objs_path`; OSMObjESRGANModel is outside the hot path)."""
from __future__ import annotations

import glob
import json
import os
import random
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils import data as data

from ..registry import DATASET_REGISTRY


def read_png(path: str) -> torch.Tensor:
    """uint8 [C, H, W] with the file's own channel count (torchvision.io.read_image semantics for 8-bit PNGs)."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("L", "RGB", "RGBA", "LA"):
            im = im.convert("RGB")
        a = np.array(im)          # a writable copy (np.asarray of a PIL image is a read-only view)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))


def has_black_pixels(t: torch.Tensor) -> bool:
    """ssr/utils/data_utils.py:3-10: a pixel whose channel sum is 0 (uint8 sums are taken in a wider type, as torch.sum does)."""
    return bool((t.to(torch.int64).sum(dim=0) == 0).any())


class TileWeightSampler(data.WeightedRandomSampler):
    """CustomWeightedRandomSampler (:17-32): np.random.choice over the normalised weights, so that more than 2^24 samples work."""

    def __iter__(self):
        w = self.weights.numpy()
        picks = np.random.choice(len(w), size=self.num_samples, p=w / w.sum(), replace=self.replacement)
        return iter(picks.tolist())


@DATASET_REGISTRY.register()
class S2NAIPDataset(data.Dataset):
    def __init__(self, opt: Dict):
        super().__init__()
        self.opt = opt
        self.split = opt["phase"]
        train = self.split == "train"
        self.rand_crop = bool(opt.get("rand_crop", False))
        self.n_s2_images = int(opt["n_s2_images"])
        self.scale = int(opt["scale"])
        self.use_3d = bool(opt.get("use_3d", False))
        self.old_naip_path = opt.get("old_naip_path")
        if opt.get("osm_objs_path"):
            raise NotImplementedError("datasets.*.osm_objs_path: the OSM-object model is outside the MI355X hot path")
        bands = list(opt.get("s2_bands", ["tci"]))
        bands.insert(0, bands.pop(bands.index("tci")))           # tci first (:73-75)
        self.s2_bands = bands
        self.s2_path, self.naip_path = opt["sentinel2_path"], opt["naip_path"]
        if not (os.path.exists(self.s2_path) and os.path.exists(self.naip_path)):
            raise Exception("Please make sure the paths to the data directories are correct.")
        old_by_chip: Dict[str, List[str]] = {}
        if self.old_naip_path is not None:
            for p in glob.glob(self.old_naip_path + "/**/*.png", recursive=True):
                old_by_chip.setdefault(os.path.basename(p)[:-4], []).append(p)
        naip = glob.glob(self.naip_path + "/**/*.png", recursive=True)
        if "train_samples" in opt and train:
            naip = random.sample(naip, opt["train_samples"])
        self.naip_chips = naip
        self.datapoints = []
        for n in naip:
            chip = n.split("/")[-2]
            s2 = [os.path.join(self.s2_path, chip, b + ".png") for b in bands]
            self.datapoints.append([n, s2, chip] + ([old_by_chip[chip][0]] if self.old_naip_path else []))
        self.data_len = len(self.datapoints)

    def get_tile_weight_sampler(self, tile_weights: Dict[str, float]):
        """:134-152: weight of a datapoint = tile_weights[<naip file stem>], 1 when absent."""
        weights = [tile_weights.get(os.path.basename(dp[0])[:-4], 1) for dp in self.datapoints]
        return TileWeightSampler(weights, len(self.datapoints))

    def _load_s2(self, s2_paths: List[str]) -> torch.Tensor:
        """[T, C_total, 32, 32]: every band file is a [T*32, 32, c] image; a missing band contributes zeros."""
        parts = []
        for p in s2_paths:
            if not os.path.exists(p):
                parts.append(torch.zeros((self.n_s2_images, 3 if "tci" in p else 1, 32, 32), dtype=torch.uint8))
            else:
                img = read_png(p)
                parts.append(img.reshape(img.shape[0], -1, 32, 32).permute(1, 0, 2, 3))
        return torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]

    def __getitem__(self, index: int):
        skipped = 0
        while True:
            index += skipped          # the reference adds the running count of skips each time (:163)
            if index >= self.data_len:
                index = 0
            dp = self.datapoints[index]
            naip_path, s2_paths, chip = dp[0], dp[1], dp[2]
            hr = read_png(naip_path)
            if has_black_pixels(hr):
                skipped += 1
                continue
            try:
                s2 = self._load_s2(s2_paths)
            except Exception:         # a few unreadable Sentinel-2 files exist in the published set (:197-199)
                skipped += 1
                continue
            if s2.shape[0] < self.n_s2_images:
                skipped += 1
                continue
            black = [has_black_pixels(frame[:3]) for frame in s2]
            goods = [i for i, b in enumerate(black) if not b]
            bads = [i for i, b in enumerate(black) if b]
            if len(goods) >= self.n_s2_images:
                picks = random.sample(goods, self.n_s2_images)
            else:
                picks = goods + random.sample(bads, self.n_s2_images - len(goods))
            lr = s2[torch.as_tensor(picks)]
            if self.rand_crop:        # random square crop of side 24..32 (x4 on the HR side), nearest resize back (:227-233)
                side = random.randint(24, 32)
                lr = F.interpolate(lr[:, :, :side, :side], (32, 32))
                hr = F.interpolate(hr[:, :4 * side, :4 * side].unsqueeze(0), (128, 128)).squeeze(0)
            if not self.use_3d:
                lr = lr.reshape(-1, 32, 32)
            out = {"hr": hr, "lr": lr, "Index": index, "Phase": self.split, "Chip": chip}
            if self.old_naip_path is not None:
                out["old_hr"] = read_png(dp[3])
            return out

    def __len__(self):
        return self.data_len


def build_train_loader(dataset: S2NAIPDataset, opt: Dict, tile_weights_path: str = None, rank: int = 0, world: int = 1,
                       seed: int = None):
    """The loader ssr/train.py builds (create_train_val_dataloader): batch_size_per_gpu samples per rank, num_worker_per_gpu worker
    processes, the tile-weight sampler when `tile_weights` is given; pinned uint8 batches so that feed_data's upload is one
    asynchronous copy.  `seed` is the option file's top-level `manual_seed` (`opt` here is the dataset sub-dict, which does not
    carry it): None = unseeded, as in the reference (options.py:79-82 seeds rank r with manual_seed + r only when a seed is set)."""
    sampler = None
    if tile_weights_path or opt.get("tile_weights"):
        with open(tile_weights_path or opt["tile_weights"]) as f:
            sampler = dataset.get_tile_weight_sampler(json.load(f))
        if seed is not None:
            # the weighted sampler draws from numpy's global generator: every rank its own stream (options.py:79-82 seeds
            # rank r with manual_seed + r), so that the ranks do not draw the same tiles; without a configured seed numpy stays
            # unseeded at every world size (each process then has its own entropy)
            np.random.seed((int(seed) + rank) % (2 ** 32))
    elif world > 1:
        # data parallel: every rank iterates its own share of the dataset (BasicSR builds an EnlargedSampler(train_set, world,
        # rank, ratio); DistributedSampler is the same partition at ratio 1), reshuffled per epoch through set_epoch()
        sampler = data.distributed.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=bool(opt.get("use_shuffle", False)),
                                                      seed=int(seed) if seed is not None else 0, drop_last=True)
    return data.DataLoader(dataset, batch_size=int(opt["batch_size_per_gpu"]), shuffle=bool(opt.get("use_shuffle", False)) and sampler is None,
                           sampler=sampler, num_workers=int(opt.get("num_worker_per_gpu", 0)), drop_last=True,
                           pin_memory=torch.cuda.is_available(), persistent_workers=int(opt.get("num_worker_per_gpu", 0)) > 0)
