"""CPU: host-side pieces of the inference / training drivers that round 4 added or rewrote - the PNG writer and the
shared-memory worker pool of satlas_super_resolution_amd/png_io.py (used by infer_grid.py / infer.py, which mirror
/root/reference/ssr/infer_grid.py:46-85 and ssr/infer.py:40-62), and the resume rules of train.py (BasicSR's
load_resume_state + check_resume, which the reference's ssr/train.py:64-65 calls).  No GPU, no oracle."""
import os

import numpy as np
import pytest
import torch

from satlas_super_resolution_amd import png_io


@pytest.mark.parametrize("shape", [(128, 128, 3), (1, 1, 3), (7, 13, 3), (512, 512, 3)])
def test_own_png_writer_is_read_back_bit_exact_by_pillow(tmp_path, shape):
    from PIL import Image
    rng = np.random.RandomState(sum(shape))
    img = rng.randint(0, 256, shape).astype(np.uint8)
    blob = png_io.encode_png(img)
    assert blob[:8] == b"\x89PNG\r\n\x1a\n"
    p = tmp_path / "a.png"
    p.write_bytes(blob)
    with Image.open(p) as im:
        assert im.mode == "RGB" and im.size == (shape[1], shape[0])
        assert np.array_equal(np.array(im), img)
    assert np.array_equal(png_io.read_png(str(p)), img)


def test_mosaic_places_cell_ij_at_row_i_column_j_and_takes_the_first_frame_of_s2_stacks():
    """infer_utils.stitch (/root/reference/ssr/utils/infer_utils.py:41-60)"""
    cells = [np.full((8, 8, 3), 16 * i + j, np.uint8) for i in range(16) for j in range(16)]
    m = png_io.mosaic(cells, 128, 16, False)
    assert m.shape == (128, 128, 3)
    for i in (0, 5, 15):
        for j in (0, 9, 15):
            assert (m[8 * i:8 * i + 8, 8 * j:8 * j + 8] == 16 * i + j).all()
    stacks = [np.concatenate([np.full((32, 32, 3), k, np.uint8), np.full((64, 32, 3), 255, np.uint8)]) for k in range(256)]
    s2 = png_io.mosaic(stacks, 512, 16, True)
    assert s2.shape == (512, 512, 3) and (s2[32 * 3:32 * 4, 32 * 7:32 * 8] == 3 * 16 + 7).all()
    # a size that the grid does not divide: the margin stays zero, as in the reference
    odd = png_io.mosaic([np.ones((6, 6, 3), np.uint8)] * 256, 100, 16, False)
    assert odd.shape == (100, 100, 3) and (odd[:96, :96] == 1).all() and not odd[96:].any() and not odd[:, 96:].any()


def test_shared_block_outlives_its_path_while_a_view_is_alive_and_closes_twice():
    blk = png_io.ShmBlock(4096, png_io.shm_dir(4096), "t")
    blk.buf[:4] = [1, 2, 3, 4]
    view, path = blk.buf[:4], blk.path
    blk.close()
    assert not os.path.exists(path) and view.tolist() == [1, 2, 3, 4]
    blk.close()
    a, b = png_io.ShmBlock(16, png_io.shm_dir(16), "x"), png_io.ShmBlock(16, png_io.shm_dir(16), "x")
    assert a.path != b.path        # never the same path twice: long-lived workers cache their mappings by path
    a.close(); b.close()


@pytest.mark.parametrize("workers", [2, 0])
def test_worker_pool_decodes_into_and_encodes_from_a_shared_block(tmp_path, workers):
    rng = np.random.RandomState(1)
    imgs = [rng.randint(0, 256, (96, 32, 3)).astype(np.uint8) for _ in range(5)] + [rng.randint(0, 256, (640, 32, 3)).astype(np.uint8)]
    paths = []
    for k, im in enumerate(imgs):
        paths.append(str(tmp_path / f"in{k}.png"))
        png_io.save_png(im, paths[-1])
    slot = 96 * 32 * 3
    blk = png_io.ShmBlock(len(imgs) * slot, png_io.shm_dir(len(imgs) * slot), "in")
    out = png_io.ShmBlock(2 * 128 * 128 * 3, png_io.shm_dir(1 << 20), "out")
    try:
        with png_io.PngWorkerPool(workers, threads=2) as pool:
            shapes = pool.submit("read_into", paths, blk.path, blk.nbytes, [slot * k for k in range(len(imgs))], slot).result()
            for k in range(5):
                assert shapes[k] == (96, 32, 3)
                assert np.array_equal(blk.buf[slot * k:slot * (k + 1)].reshape(96, 32, 3), imgs[k])
            assert isinstance(shapes[5], np.ndarray) and np.array_equal(shapes[5], imgs[5])      # does not fit its slot: comes back by value
            tiles = rng.randint(0, 256, (2, 128, 128, 3)).astype(np.uint8)
            out.buf[:] = tiles.reshape(-1)
            dst = [str(tmp_path / "o" / f"{k}.png") for k in range(2)]
            n = pool.submit("save_from", out.path, out.nbytes, [(k * 128 * 128 * 3, (128, 128, 3), dst[k]) for k in range(2)]).result()
            assert n == 2 and all(np.array_equal(png_io.read_png(dst[k]), tiles[k]) for k in range(2))
            # a failing task raises in the caller with the worker's traceback; the pool stays usable
            with pytest.raises(Exception) as e:
                pool.submit("read_many", [str(tmp_path / "missing.png")]).result()
            assert "missing.png" in str(e.value)
            assert pool.submit("read_many", paths[:1]).result()[0].shape == (96, 32, 3)
    finally:
        blk.close(); out.close()


def test_one_shot_blocks_are_not_retained_and_the_mapping_cache_is_bounded(tmp_path):
    """A tile's mosaic goes to an encoder through a block of its own that the driver unlinks when the file is written: the worker
    must not keep a mapping of it (an unlinked tmpfs file stays allocated while mapped - 12.6 MB per tile and worker).  The cache of
    the fixed blocks is bounded and closes what it evicts."""
    img = np.random.RandomState(2).randint(0, 256, (64, 64, 3)).astype(np.uint8)
    n0 = len(png_io._MAPS)
    for k in range(3):
        blk = png_io.ShmBlock(img.nbytes, png_io.shm_dir(img.nbytes), "mosaic")
        blk.buf[:] = img.reshape(-1)
        dst = str(tmp_path / f"m{k}.png")
        assert png_io.save_from(blk.path, blk.nbytes, [(0, img.shape, dst)], True) == 1      # transient: mapped, encoded, unmapped
        assert blk.path not in png_io._MAPS and len(png_io._MAPS) == n0
        blk.close()
        assert np.array_equal(png_io.read_png(dst), img)
    blks = [png_io.ShmBlock(64, png_io.shm_dir(64), "ring") for _ in range(png_io._MAPS_MAX + 3)]
    try:
        for b in blks:
            png_io._map(b.path, b.nbytes)[0] = 7
        assert len(png_io._MAPS) == png_io._MAPS_MAX and blks[0].path not in png_io._MAPS and blks[-1].path in png_io._MAPS
        assert all(int(b.buf[0]) == 7 for b in blks)
        png_io._map(blks[3].path, 64)                       # a hit moves the entry to the young end
        assert list(png_io._MAPS)[-1] == blks[3].path
        # an evicted / transient mapping is really CLOSED (mmap.closed), not merely forgotten: the entry's own view is dropped first
        e = png_io._open_map(blks[0].path, 64)
        mm = e[1]
        assert png_io._close_map(e) and mm.closed and e[0] is None
        e = png_io._open_map(blks[0].path, 64)
        held = e[0]                                         # a caller that still holds a view: the close is refused, and reported
        assert not png_io._close_map(e) and not e[1].closed
        del held
        assert png_io._close_map(e) and e[1].closed
    finally:
        for b in blks:
            e = png_io._MAPS.pop(b.path, None)
            if e is not None:
                assert png_io._close_map(e) and e[1].closed
            b.close()


def test_a_dead_worker_fails_the_task_instead_of_hanging(tmp_path):
    with png_io.PngWorkerPool(1) as pool:
        pool.procs[0].kill()
        pool.procs[0].wait()
        with pytest.raises(BaseException):
            pool.submit("read_many", []).result(timeout=30)


def test_host_cores_is_positive_and_not_above_the_affinity_mask():
    n = png_io.host_cores()
    assert 1 <= n <= (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count())


# ---- resume rules (train.py: BasicSR load_resume_state / check_resume) ----
def _resume_tree(tmp_path, iters=(100, 200), nets=("g", "d")):
    models, states = tmp_path / "models", tmp_path / "training_states"
    models.mkdir(); states.mkdir()
    for it in iters:
        torch.save({"iter": it, "epoch": 0, "optimizers": [], "schedulers": []}, states / f"{it}.state")
        for n in nets:
            torch.save({"params": {}}, models / f"net_{n}_{it}.pth")
    return {"path": {"models": str(models), "training_states": str(states), "pretrain_network_g": "weights/esrgan.pth",
                     "pretrain_network_d": None}}


def test_auto_resume_takes_the_newest_state_overrides_resume_state_and_redirects_both_networks(tmp_path):
    from satlas_super_resolution_amd.train import resolve_resume
    opt = _resume_tree(tmp_path)
    opt["path"]["resume_state"] = os.path.join(opt["path"]["training_states"], "100.state")
    log = []
    st = resolve_resume(opt, auto_resume=True, log=log.append)
    assert st["iter"] == 200
    assert opt["path"]["pretrain_network_g"].endswith("net_g_200.pth") and opt["path"]["pretrain_network_d"].endswith("net_d_200.pth")
    assert any("pretrain_network_g is redirected" in m for m in log)
    # without the flag the option file's state is used
    (tmp_path / "b").mkdir()
    opt2 = _resume_tree(tmp_path / "b")
    opt2["path"]["resume_state"] = os.path.join(opt2["path"]["training_states"], "100.state")
    assert resolve_resume(opt2, auto_resume=False, log=log.append)["iter"] == 100
    assert opt2["path"]["pretrain_network_g"].endswith("net_g_100.pth")


def test_resume_without_the_matching_network_file_is_an_error_unless_the_network_is_ignored(tmp_path):
    from satlas_super_resolution_amd.train import resolve_resume
    opt = _resume_tree(tmp_path, iters=(300,), nets=("g",))
    opt["path"]["resume_state"] = os.path.join(opt["path"]["training_states"], "300.state")
    with pytest.raises(FileNotFoundError) as e:
        resolve_resume(opt, log=lambda m: None)
    assert "net_d_300.pth" in str(e.value) and "ignore_resume_networks" in str(e.value)
    opt["path"]["ignore_resume_networks"] = ["network_d"]
    st = resolve_resume(opt, log=lambda m: None)
    assert st["iter"] == 300 and opt["path"]["pretrain_network_d"] is None and opt["path"]["pretrain_network_g"].endswith("net_g_300.pth")


def test_resume_resets_param_key_params_ema_to_params(tmp_path):
    """BasicSR's check_resume ends by resetting every path.param_key_* that reads 'params_ema' to 'params': the shipped option
    files say `param_key_g: params_ema` (esrgan_s2naip_urban.yml:86-92) and a resume must load the TRAINED weights - the restored
    Adam moments and counters belong to them - while the EMA comes from the checkpoint's 'params_ema' entry."""
    from satlas_super_resolution_amd.train import resolve_resume
    opt = _resume_tree(tmp_path, iters=(100,))
    opt["path"].update(resume_state=os.path.join(opt["path"]["training_states"], "100.state"), param_key_g="params_ema",
                       param_key_d="params", strict_load_g=True)
    log = []
    assert resolve_resume(opt, log=log.append)["iter"] == 100
    assert opt["path"]["param_key_g"] == "params" and opt["path"]["param_key_d"] == "params"
    assert any("param_key_g is reset" in m for m in log)
    # a fresh run (no state) keeps the option file's key: fine-tuning from published EMA weights
    opt2 = {"path": {"models": str(tmp_path / "m"), "training_states": str(tmp_path / "s"), "pretrain_network_g": "w.pth",
                     "param_key_g": "params_ema"}}
    assert resolve_resume(opt2, auto_resume=True) is None and opt2["path"]["param_key_g"] == "params_ema"


def test_no_state_anywhere_means_a_fresh_run(tmp_path):
    from satlas_super_resolution_amd.train import resolve_resume
    opt = {"path": {"models": str(tmp_path / "m"), "training_states": str(tmp_path / "s"), "pretrain_network_g": "w.pth"}}
    assert resolve_resume(opt, auto_resume=True) is None and opt["path"]["pretrain_network_g"] == "w.pth"
