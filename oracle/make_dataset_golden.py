"""TEST INFRASTRUCTURE — a miniature S2-NAIP dataset on disk (tests/golden/s2naip_mini/, synthetic images) and the samples the
UNMODIFIED reference dataset class (/root/reference/ssr/data/s2-naip_dataset.py:35-246) returns for it under fixed `random`
seeds (tests/golden/s2naip_samples.pt).  The reference needs `torchvision.io.read_image`, which is not installed here: a
Pillow-backed stand-in with the same contract (uint8 [C,H,W], the file's own channels) is injected — a dependency stub, the
reference's own code runs unmodified.  Run in the build container:  python oracle/make_dataset_golden.py"""
import importlib
import os
import random
import shutil
import sys
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

MINI = os.path.join(ROOT, "tests", "golden", "s2naip_mini")
CHIPS = ["100_200", "100_201", "101_200", "101_201", "102_200"]


def smooth(rng, h, w, c, lo=1):
    """compressible synthetic imagery: a few gradients and rectangles, values >= lo (no accidental black pixels)"""
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, c), np.float32)
    for k in range(c):
        img[..., k] = 60 + 50 * np.sin(yy / rng.uniform(5, 20) + rng.uniform(0, 6)) + 50 * np.cos(xx / rng.uniform(5, 20))
    for _ in range(4):
        y0, x0 = rng.randint(0, h - 4), rng.randint(0, w - 4)
        img[y0:y0 + rng.randint(2, h // 3), x0:x0 + rng.randint(2, w // 3)] += rng.uniform(-40, 60, size=c)
    return np.clip(img, lo, 255).astype(np.uint8)


def build_mini():
    shutil.rmtree(MINI, ignore_errors=True)
    rng = np.random.RandomState(11)
    for ci, chip in enumerate(CHIPS):
        os.makedirs(os.path.join(MINI, "naip", chip), exist_ok=True)
        os.makedirs(os.path.join(MINI, "sentinel2", chip), exist_ok=True)
        os.makedirs(os.path.join(MINI, "old_naip", "2017"), exist_ok=True)
        hr = smooth(rng, 128, 128, 3)
        if chip == "101_200":
            hr[40, 50] = 0                                   # a black pixel: the datapoint must be skipped
        Image.fromarray(hr).save(os.path.join(MINI, "naip", chip, chip + ".png"))
        Image.fromarray(smooth(rng, 128, 128, 3)).save(os.path.join(MINI, "old_naip", "2017", chip + ".png"))
        T = 6 if chip == "102_200" else 10                   # too few frames for n_s2_images = 8: skipped
        tci = smooth(rng, T * 32, 32, 3)
        for t in ([1, 4] if chip == "100_200" else [0, 2, 3, 5] if chip == "100_201" else []):
            tci[t * 32 + 3, 7] = 0                           # frames with a black pixel
        Image.fromarray(tci).save(os.path.join(MINI, "sentinel2", chip, "tci.png"))
        if chip != "101_201":                                # one chip lacks the extra band: zeros are substituted
            Image.fromarray(smooth(rng, T * 32, 32, 1)[..., 0]).save(os.path.join(MINI, "sentinel2", chip, "b08.png"))


def load_reference_dataset():
    ref_shim._install_stubs()
    ref_shim.load_reference_archs()
    tv = sys.modules["torchvision"]
    io = types.ModuleType("torchvision.io")

    def read_image(path):
        a = np.array(Image.open(path))
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    io.read_image = read_image
    tv.io = io
    sys.modules["torchvision.io"] = io
    for sub in ("data", "utils"):
        name = f"ssr.{sub}"
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(ref_shim.REFERENCE_ROOT, "ssr", sub)]
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ssr.data.s2naip_dataset_ref",
                                                  os.path.join(ref_shim.REFERENCE_ROOT, "ssr", "data", "s2-naip_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.S2NAIPDataset


CONFIGS = {
    "plain": dict(n_s2_images=8),
    "rand_crop": dict(n_s2_images=8, rand_crop=True),
    "use_3d": dict(n_s2_images=4, use_3d=True),
    "bands": dict(n_s2_images=8, s2_bands=["b08", "tci"]),
    "old_hr": dict(n_s2_images=8, old_naip_path=os.path.join(MINI, "old_naip")),
}


def base_opt(**over):
    opt = {"phase": "train", "scale": 4, "name": "mini", "sentinel2_path": os.path.join(MINI, "sentinel2"),
           "naip_path": os.path.join(MINI, "naip")}
    opt.update(over)
    return opt


def main():
    build_mini()
    DS = load_reference_dataset()
    out = {}
    for cname, over in CONFIGS.items():
        ds = DS(base_opt(**{k: (list(v) if isinstance(v, list) else v) for k, v in over.items()}))
        by_chip = {dp[2]: i for i, dp in enumerate(ds.datapoints)}
        # the valid chips: their own sample comes back.  (With an extra band, 101_201 — which lacks the band file — is skipped by
        # the reference: its zero substitute has n_s2_images frames, the tci file 10, and the concatenation raises.)
        for chip in (("100_200", "100_201") if cname == "bands" else ("100_200", "100_201", "101_201")):
            random.seed(1000 + len(out))
            s = ds[by_chip[chip]]
            assert s["Chip"] == chip
            out[(cname, chip)] = {"seed": 1000 + len(out), **{k: v for k, v in s.items() if k != "Index"}}
    torch.save(out, os.path.join(ROOT, "tests", "golden", "s2naip_samples.pt"))
    n = sum(len(f) for _, _, f in os.walk(MINI))
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(MINI) for f in fs)
    print(f"wrote {len(out)} samples; mini dataset: {n} files, {size / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
