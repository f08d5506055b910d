"""GPU: this package's inference drivers against what the UNMODIFIED reference scripts wrote
(/root/reference/ssr/infer_grid.py:15-85 and ssr/infer.py:14-67, executed as __main__ on the CPU by oracle/make_infer_golden.py ->
tests/golden/infer_scripts.pt): same procedurally generated input tree, same option files, same weights -> same file layout and
names, same stitched mosaics, pixel values within one uint8 level (the scripts truncate `output * 255`, so a value within fp32
rounding of an integer may land on either side)."""
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _setup(tmp_path, which):
    from oracle import make_infer_golden as M
    M.write_inputs(str(tmp_path))
    M.write_weights(str(tmp_path / "w.pth"))
    txt = M.option_text(str(tmp_path / which) + "/", str(tmp_path / ("out_" + which)) + "/", str(tmp_path / "w.pth"))
    return M, yaml.safe_load(txt)


def _levels(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return int(d.max()), float((d > 0).mean())


@pytest.mark.parametrize("compute_dtype", ["fp32h", "fp32x3", "fp32"])      # fp32h: the driver's default arithmetic
def test_infer_grid_driver_matches_the_unmodified_reference_script(tmp_path, compute_dtype):
    from satlas_super_resolution_amd.infer_grid import run_infer_grid
    fx = load_golden("infer_scripts")
    M, opt = _setup(tmp_path, "grid")
    opt["compute_dtype"] = compute_dtype
    res = run_infer_grid(opt)
    assert (res["chunks"], res["tiles_stitched"]) == (259, 1)
    tree = M.read_tree(str(tmp_path / "out_grid"))
    assert sorted(tree) == fx["grid_files"]                                   # names and layout, incl. the tile that cannot be stitched
    assert {k: tuple(v.shape) for k, v in tree.items()} == fx["grid_shapes"]
    assert M.digest(tree["t0/stitched_s2.png"]) == fx["grid_stitched_s2_sha256"]          # input mosaic: exact
    worst, frac = 0, 0.0
    for k, ref in fx["grid_chunks"].items():
        mx, fr = _levels(tree[k], ref.numpy())
        worst, frac = max(worst, mx), max(frac, fr)
    mx, fr = _levels(tree["t0/stitched_sr.png"][::8, ::8], fx["grid_stitched_sr_sub8"].numpy())
    mx2, _ = _levels(tree["t0/stitched_sr.png"][640], fx["grid_stitched_sr_row640"].numpy())
    exact = sum(M.digest(v) == fx["grid_chunk_sha256"][k] for k, v in tree.items() if "stitched" not in k)
    print(f"[{compute_dtype}] chunks identical to the reference script's: {exact} of 259; worst difference {max(worst, mx, mx2)} level(s), "
          f"fraction of differing samples <= {max(frac, fr):.2e}")
    assert max(worst, mx, mx2) <= 1 and max(frac, fr) <= 2e-3
    # the stitched super-resolution is the mosaic of this run's own chunks (reference layout: cell (i, j) at rows 128 i, columns 128 j)
    assert np.array_equal(tree["t0/stitched_sr.png"][128 * 3:128 * 4, 128 * 5:128 * 6], tree["t0/3_5.png"])
    assert float(tree["t0/stitched_sr.png"].std()) > 5           # real images, not a constant


def test_infer_driver_matches_the_unmodified_reference_script(tmp_path):
    from satlas_super_resolution_amd.infer import run_infer
    fx = load_golden("infer_scripts")
    M, opt = _setup(tmp_path, "single")
    assert run_infer(opt) == {"images": 5}
    tree = M.read_tree(str(tmp_path / "out_single"))
    assert sorted(tree) == fx["single_files"]
    for k in range(5):
        lr, sr = tree[f"{k}/lr.png"], tree[f"{k}/sr.png"]
        ref = fx["single_pairs"][M.digest(lr)].numpy()           # the reference's {i} follows glob's order: match by the low-res image
        mx, fr = _levels(sr, ref)
        assert sr.shape == (128, 128, 3) and mx <= 1 and fr <= 2e-3, (k, mx, fr)


@pytest.mark.parametrize("world", [2, 8])
def test_infer_grid_n_ranks_write_their_shares_and_rank0_stitches_from_disk(tmp_path, world):
    """`python -m torch.distributed.run --nproc-per-node N -m satlas_super_resolution_amd.infer_grid -opt ...` as N (2; 8 = the
    reference's node, README.md:159) processes on this GPU (gloo standing in for RCCL; the data path has no collective, only the
    barrier in front of the stitch): rank r runs and writes chunks r, r + N, ..., rank 0 then builds both mosaics of every complete
    tile by RE-READING the chunk files of all ranks.  With eight ranks the PNG workers come from the per-rank budget
    (host cores / ranks - 1, at least one) and the run's wall time is printed: the host-bound share of whole-node inference.
    (the rest of the original note:) rank 0 re-reads the files of both
    ranks (the reference's stitch(), /root/reference/ssr/utils/infer_utils.py:41-60).  Frame selection draws from `random` per
    processed chunk (infer_utils.py:22-30), so a sharded run picks other frame orders than a serial one: layout, names, shapes and the
    input mosaic are compared with the reference script's run, the super-resolved mosaic with this run's own chunk files."""
    import socket
    import subprocess
    import sys
    fx = load_golden("infer_scripts")
    M, opt = _setup(tmp_path, "grid")
    opt["compute_dtype"] = "fp32x3"
    if world == 2:
        opt["io_workers"] = 2                     # (8 ranks: the default budget, cores / ranks - 1)
    with open(tmp_path / "opt.yml", "w") as f:
        yaml.safe_dump(opt, f)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   SSR_DIST_BACKEND="gloo", PYTHONPATH=root)
        procs.append(subprocess.Popen([sys.executable, "-m", "satlas_super_resolution_amd.infer_grid", "-opt", str(tmp_path / "opt.yml")],
                                      cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import re
    for r in range(world):                       # 259 chunks dealt round-robin
        assert f"'chunks': {len(range(r, 259, world))}" in outs[r], outs[r]
        assert f"'tiles_stitched': {1 if r == 0 else 0}" in outs[r], outs[r]
    secs = [float(re.search(r"'seconds': ([0-9.]+)", o).group(1)) for o in outs]
    wk = [int(re.search(r"'io_workers': (\d+)", o).group(1)) for o in outs]
    print(f"infer_grid, {world} ranks on one host / one GPU: io_workers per rank {wk}, slowest rank {max(secs):.2f} s for 259 chunks "
          f"= {259 / 256 / max(secs):.2f} tiles/s whole-job (model load and pool start-up included)")
    tree = M.read_tree(str(tmp_path / "out_grid"))
    assert sorted(tree) == fx["grid_files"]
    assert {k: tuple(v.shape) for k, v in tree.items()} == fx["grid_shapes"]
    assert M.digest(tree["t0/stitched_s2.png"]) == fx["grid_stitched_s2_sha256"]
    sr = tree["t0/stitched_sr.png"]
    for i in range(16):
        for j in range(16):               # every cell of the mosaic is the chunk file some rank wrote
            assert np.array_equal(sr[128 * i:128 * (i + 1), 128 * j:128 * (j + 1)], tree[f"t0/{i}_{j}.png"]), (i, j)
    assert float(sr.std()) > 5
