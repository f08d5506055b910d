"""TEST INFRASTRUCTURE — runs the UNMODIFIED reference inference scripts

    /root/reference/ssr/infer_grid.py   (whole-tile inference: per-chunk PNGs + stitched_sr.png / stitched_s2.png)
    /root/reference/ssr/infer.py        (per-image inference: {i}/lr.png + {i}/sr.png)

as `__main__` (runpy) on a procedurally generated miniature input tree, on the CPU, with the reference's own SSR_RRDBNet, and
stores what they wrote as tests/golden/infer_scripts.pt.  The GPU tests (tests/test_gpu_infer_scripts.py) run this package's
drivers (satlas_super_resolution_amd/infer_grid.py, infer.py) on the SAME inputs, option files and weights and compare file
layout, names and pixel values.

What is stood in for (packages that are not installed here; none of it is arithmetic of the hot path):
  skimage.io.imread / imsave      -> Pillow (PNG is lossless; the arrays, not the file bytes, are compared)
  torchvision                     -> empty module (infer_grid.py imports it and never uses it)
  basicsr.utils.set_random_seed, basicsr.utils.dist_util   -> import-time names of ssr/utils/options.py
  ssr.archs.highresnet_arch / srcnn_arch   -> empty classes (ssr/utils/model_utils.py imports them; other model types)
  torch.device('cuda')            -> the scripts hard-code it (infer_grid.py:21, infer.py:19); mapped to the CPU for this run

Inputs carry ONE Sentinel-2 frame per chunk (n_lr_images = 1), so that `format_s2naip_data`'s random frame choice cannot depend
on the order in which glob lists the files (the frame choice itself is pinned by tests/golden/infer_utils.pt).

    python -m oracle.make_infer_golden          (build container only: needs /root/reference)
"""
import hashlib
import os
import random
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

from oracle.ref_shim import REFERENCE_ROOT, load_reference_archs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
G_KW = dict(num_feat=16, num_block=1, num_grow_ch=8)
SEED = 4242


# ------------------------------------------------------------------ shared with the GPU test: inputs, weights, option files
def chunk_image(tile: int, i: int, j: int) -> np.ndarray:
    """one [1*32, 32, 3] uint8 Sentinel-2 chunk: smooth field + deterministic noise, values in [1, 255] (no black samples)"""
    yy, xx = np.mgrid[0:32, 0:32].astype(np.float64)
    rng = np.random.RandomState(SEED + 1000 * tile + 16 * i + j)
    base = np.stack([96 + 64 * np.sin((yy + 32 * i) / 23.0 + c) * np.cos((xx + 32 * j) / 17.0 - c) for c in range(3)], -1)
    img = base + rng.randint(-20, 21, size=(32, 32, 3))
    return np.clip(np.rint(img), 1, 255).astype(np.uint8)


def write_inputs(root: str):
    """{root}/grid/t0/{i}_{j}.png (a complete 16 x 16 tile), {root}/grid/t1/0_{j}.png (3 chunks: too few to stitch),
    {root}/single/a/{k}.png (5 images for infer.py)"""
    from PIL import Image
    for tile, cells in (("t0", [(i, j) for i in range(16) for j in range(16)]), ("t1", [(0, 0), (0, 1), (0, 2)])):
        d = os.path.join(root, "grid", tile)
        os.makedirs(d, exist_ok=True)
        for (i, j) in cells:
            Image.fromarray(chunk_image(int(tile[1]), i, j)).save(os.path.join(d, f"{i}_{j}.png"))
    d = os.path.join(root, "single", "a")
    os.makedirs(d, exist_ok=True)
    for k in range(5):
        Image.fromarray(chunk_image(7, k, 3)).save(os.path.join(d, f"{k}.png"))


def write_weights(path: str):
    from oracle import esrgan_oracle as O
    sd = O.generator_init(num_in_ch=3, num_out_ch=3, scale=4, seed=SEED, **G_KW)
    # default init leaves the output near 0.0: give conv_last a bias so that the uint8 images are not all black
    sd["conv_last.bias"] = torch.full_like(sd["conv_last.bias"], 0.45)
    sd["conv_last.weight"] = sd["conv_last.weight"] * 8
    torch.save({"params_ema": sd, "params": sd}, path)
    return sd


def option_text(data_dir: str, save_path: str, weights: str) -> str:
    return f"""name: infer_fixture
model_type: SSRESRGANModel
scale: 4
num_gpu: auto
manual_seed: 0
data_dir: {data_dir}
n_lr_images: 1
save_path: {save_path}
network_g:
  type: SSR_RRDBNet
  num_in_ch: 3
  num_out_ch: 3
  num_feat: {G_KW['num_feat']}
  num_block: {G_KW['num_block']}
  num_grow_ch: {G_KW['num_grow_ch']}
path:
  pretrain_network_g: {weights}
  param_key_g: params_ema
  strict_load_g: true
"""


def read_tree(root: str):
    """{relative path: uint8 array} of every PNG under root"""
    from PIL import Image
    out = {}
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".png"):
                p = os.path.join(dp, f)
                out[os.path.relpath(p, root)] = np.asarray(Image.open(p).convert("RGB")).copy()
    return out


def digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ the stand-ins and the run
def _install():
    load_reference_archs()                      # basicsr stand-ins + ssr / ssr.archs packages + the real rrdbnet_arch
    from PIL import Image
    sk, skio = types.ModuleType("skimage"), types.ModuleType("skimage.io")
    skio.imread = lambda path: np.asarray(Image.open(path)).copy()
    skio.imsave = lambda path, arr, check_contrast=False: Image.fromarray(np.asarray(arr)).save(path)
    sk.io = skio
    sys.modules.update({"skimage": sk, "skimage.io": skio})
    u = sys.modules["basicsr.utils"]
    u.set_random_seed = lambda seed: None
    du = types.ModuleType("basicsr.utils.dist_util")
    du.get_dist_info = lambda: (0, 1)
    du.init_dist = lambda *a, **k: None
    du.master_only = lambda f: f
    u.dist_util = du
    sys.modules["basicsr.utils.dist_util"] = du
    for name, cls in (("ssr.archs.highresnet_arch", "HighResNet"), ("ssr.archs.srcnn_arch", "SRCNN")):
        m = types.ModuleType(name)
        setattr(m, cls, type(cls, (), {}))
        sys.modules[name] = m


def run_reference_script(script: str, opt_path: str):
    _install()
    real_device = torch.device
    argv = sys.argv

    class _Dev:                                   # torch.device('cuda') -> cpu for the duration of the script
        def __new__(cls, *a, **k):
            if a and a[0] == "cuda":
                return real_device("cpu")
            return real_device(*a, **k)
    torch.device = _Dev
    sys.argv = [script, "-opt", opt_path]
    try:
        random.seed(0)
        with torch.no_grad():
            runpy.run_path(os.path.join(REFERENCE_ROOT, "ssr", script), run_name="__main__")
    finally:
        torch.device = real_device
        sys.argv = argv


def main():
    tmp = tempfile.mkdtemp(prefix="infer_golden_")
    try:
        write_inputs(tmp)
        write_weights(os.path.join(tmp, "w.pth"))
        fx = {"g_kwargs": dict(num_in_ch=3, num_out_ch=3, scale=4, **G_KW), "seed": SEED}
        # ---- infer_grid.py
        opt = os.path.join(tmp, "grid.yml")
        open(opt, "w").write(option_text(os.path.join(tmp, "grid") + "/", os.path.join(tmp, "out_grid") + "/", os.path.join(tmp, "w.pth")))
        run_reference_script("infer_grid.py", opt)
        tree = read_tree(os.path.join(tmp, "out_grid"))
        keep = ["t0/0_0.png", "t0/3_5.png", "t0/7_15.png", "t0/15_0.png", "t0/15_15.png", "t1/0_2.png"]
        fx["grid_files"] = sorted(tree)
        fx["grid_shapes"] = {k: tuple(v.shape) for k, v in tree.items()}
        fx["grid_chunks"] = {k: torch.from_numpy(tree[k]) for k in keep}
        fx["grid_stitched_sr_sub8"] = torch.from_numpy(tree["t0/stitched_sr.png"][::8, ::8].copy())
        fx["grid_stitched_sr_row640"] = torch.from_numpy(tree["t0/stitched_sr.png"][640].copy())
        fx["grid_stitched_s2_sha256"] = digest(tree["t0/stitched_s2.png"])
        fx["grid_chunk_sha256"] = {k: digest(v) for k, v in tree.items() if "stitched" not in k}
        # ---- infer.py
        opt = os.path.join(tmp, "single.yml")
        open(opt, "w").write(option_text(os.path.join(tmp, "single") + "/", os.path.join(tmp, "out_single") + "/", os.path.join(tmp, "w.pth")))
        run_reference_script("infer.py", opt)
        tree = read_tree(os.path.join(tmp, "out_single"))
        fx["single_files"] = sorted(tree)
        # {i} is the position in glob's listing: key the pairs by the low-res image they belong to
        fx["single_pairs"] = {digest(tree[f"{k}/lr.png"]): torch.from_numpy(tree[f"{k}/sr.png"]) for k in range(5)}
        torch.save(fx, os.path.join(OUT, "infer_scripts.pt"))
        print("infer_scripts.pt:", len(fx["grid_files"]), "grid files,", len(fx["single_files"]), "single files;",
              "sr mean", float(fx["grid_stitched_sr_sub8"].float().mean()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
