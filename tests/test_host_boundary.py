"""CPU: the host-side mirror of the reference's plugin interface: registry names, ctor
signatures, state_dict layout (checked against the reference modules' own state_dicts stored in the
golden fixtures), YAML-style option handling."""
import inspect

import pytest
import torch

from conftest import load_golden


def test_registry_names_and_signatures():
    from satlas_super_resolution_amd import archs, models  # noqa: F401  (filename-suffix discovery)
    from satlas_super_resolution_amd.registry import ARCH_REGISTRY, MODEL_REGISTRY, build_network
    G = ARCH_REGISTRY.get("SSR_RRDBNet")
    D = ARCH_REGISTRY.get("SSR_UNetDiscriminatorSN")
    M = MODEL_REGISTRY.get("SSRESRGANModel")
    pg = list(inspect.signature(G.__init__).parameters)[1:7]
    assert pg == ["num_in_ch", "num_out_ch", "scale", "num_feat", "num_block", "num_grow_ch"]   # rrdbnet_arch.py:92
    pd = list(inspect.signature(D.__init__).parameters)[1:4]
    assert pd == ["num_in_ch", "num_feat", "skip_connection"]                                    # discriminator_arch.py:23
    assert list(inspect.signature(M.__init__).parameters)[1:] == ["opt"]
    for name in ("feed_data", "optimize_parameters", "test", "get_current_visuals", "get_current_log",
                 "update_learning_rate", "get_current_learning_rate", "save", "resume_training", "validation"):
        assert callable(getattr(M, name)), name
    # network_g of esrgan_s2naip_urban.yml:70-76 (num_in_ch 36 is legal and must be honoured)
    net = build_network({"type": "SSR_RRDBNet", "num_in_ch": 36, "num_out_ch": 3, "num_feat": 64, "num_block": 1,
                         "num_grow_ch": 32})
    assert net.conv_first.weight.shape == (64, 36, 3, 3)


@pytest.mark.parametrize("name", ["g_tiny_ragged", "g_mid_24ch", "g_scale2", "g_scale1"])
def test_generator_state_dict_layout_matches_reference(name):
    from satlas_super_resolution_amd.archs.rrdbnet_arch import SSR_RRDBNet
    fx = load_golden(name)
    net = SSR_RRDBNet(**fx["kwargs"])
    ref = fx["state_dict"]
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype, k
    net.load_state_dict(ref, strict=True)   # published checkpoints load with strict_load_g: true
    # init distributions (rrdbnet_arch.py:35): RDB convs ~ kaiming*0.1 with zero bias
    fresh = SSR_RRDBNet(**fx["kwargs"])
    w = fresh.body[0].rdb1.conv1.weight if hasattr(fresh.body, "__getitem__") else getattr(fresh.body, "0").rdb1.conv1.weight
    fan_in = w.shape[1] * 9
    assert abs(float(w.std()) - 0.1 * (2.0 / fan_in) ** 0.5) < 0.3 * 0.1 * (2.0 / fan_in) ** 0.5
    assert float(getattr(fresh.body, "0").rdb1.conv1.bias.abs().max()) == 0.0


@pytest.mark.parametrize("name", ["d_tiny", "d_in6_noskip"])
def test_discriminator_state_dict_layout_matches_reference(name):
    from satlas_super_resolution_amd.archs.discriminator_arch import SSR_UNetDiscriminatorSN
    fx = load_golden(name)
    net = SSR_UNetDiscriminatorSN(**fx["kwargs"])
    ref = fx["state_dict_before"]
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())     # 28 entries: weight/bias, weight_orig/weight_u/weight_v
    for k in ref:
        assert sd[k].shape == ref[k].shape, k
    net.load_state_dict(ref, strict=True)
    assert torch.allclose(net.conv1.weight_u.norm(), torch.tensor(1.0), atol=1e-5)


def test_generator_full_size_parameter_counts():
    """16,697,987 / 16,710,083 / 16,751,555 params for C_in = 3 / 24 / 96; D 4,376,897 (C_d=3)."""
    from satlas_super_resolution_amd import engine
    for cin, n in ((3, 16697987), (24, 16710083), (96, 16751555)):
        specs = engine.generator_specs(num_in_ch=cin)
        assert sum(s.cout * s.cin * 9 + s.cout for s in specs) == n
        assert len(specs) == 351
    d = engine.discriminator_specs(3)
    assert sum(s.cout * s.cin * s.k * s.k + (s.cout if s.bias else 0) for s in d) == 4376897


def test_model_option_handling_without_gpu():
    """Unsupported options must raise, never be silently ignored; KeyError semantics of the reference."""
    from satlas_super_resolution_amd.models.ssr_esrgan_model import _arch_kwargs
    with pytest.raises(NotImplementedError):
        _arch_kwargs({"type": "SRCNN"}, "SSR_RRDBNet")
    assert _arch_kwargs({"type": "SSR_RRDBNet", "num_in_ch": 24}, "SSR_RRDBNet") == {"num_in_ch": 24}


def test_format_s2naip_data_matches_reference_golden():
    """utils/infer_utils.format_s2naip_data against outputs of the unmodified reference function
    (/root/reference/ssr/utils/infer_utils.py:6-39; fixtures from oracle/make_golden.py:gen_infer_utils): same frames
    picked under the same `random` seed, black-pixel frames only used when needed, same /255 tensor."""
    import random
    from satlas_super_resolution_amd.utils.infer_utils import format_s2naip_data
    fx = load_golden("infer_utils")
    for case in fx["format_s2naip_data"]:
        random.seed(case["seed"])
        t, img = format_s2naip_data(case["data"].numpy(), case["n"], "cpu")
        assert t.shape == case["tensor"].shape and torch.equal(t, case["tensor"])
        assert torch.equal(torch.from_numpy(img.copy()), case["image"])


def test_stitch_and_sharded_chunk_loop():
    """stitch paste offsets (infer_utils.py:41-60: cell (i, j) -> rows i*cs, cols j*cs), the truncating uint8 conversion of
    infer_grid.py:60-64, and the rank-sharded chunk loop: two ranks' shares are disjoint, cover everything and equal the
    unsharded run."""
    import numpy as np
    from satlas_super_resolution_amd.utils import infer_utils as U
    chunks = {(i, j): np.full((8, 8, 3), 16 * i + j, np.uint8) for i in range(4) for j in range(4)}
    big = U.stitch_arrays(chunks, 32, grid_size=4)
    assert big.shape == (32, 32, 3) and big.dtype == np.uint8
    for i in range(4):
        for j in range(4):
            assert (big[8 * i:8 * i + 8, 8 * j:8 * j + 8] == 16 * i + j).all()
    s2 = {(i, j): np.concatenate([np.full((32, 32, 3), 7 + i + j, np.uint8), np.zeros((32, 32, 3), np.uint8)]) for i in range(2) for j in range(2)}
    assert (U.stitch_arrays(s2, 64, grid_size=2, sentinel2=True)[:32, 32:] == 8).all()
    y = torch.tensor([[[[-0.2, 0.5, 254.9 / 255, 1.7]]]]).repeat(1, 3, 1, 1)
    assert U.quantize_output(y)[0, 0, :, 0].tolist() == [0, 127, 254, 255]          # truncation, not rounding
    model = lambda x: x.mean(1, keepdim=True).repeat(1, 3, 1, 1).repeat_interleave(4, 2).repeat_interleave(4, 3)
    torch.manual_seed(0)
    inputs = [torch.rand(1, 24, 32, 32) for _ in range(11)]
    full = U.infer_chunks(model, inputs, batch=4)
    r0, r1 = U.infer_chunks(model, inputs, batch=3, rank=0, world=2), U.infer_chunks(model, inputs, batch=3, rank=1, world=2)
    assert sorted(r0) == list(range(0, 11, 2)) and sorted(r1) == list(range(1, 11, 2))
    for k, v in {**r0, **r1}.items():
        assert v.shape == (128, 128, 3) and (v == full[k]).all()


def test_discriminator_specs_steer_the_space_to_depth_path_by_grid_size():
    """engine.discriminator_specs(in_hw=...): the 4x4 stride-2 layers (discriminator_arch.py:31-33) take the space-to-depth
    big-tile path only on output grids of at least 24 rows; without a size hint the choice is left to the shape check."""
    from satlas_super_resolution_amd import engine
    flags = lambda hw: {s.name: s.s2d for s in engine.discriminator_specs(3, 64, in_hw=hw) if s.k == 4}
    assert flags(None) == {"conv1": None, "conv2": None, "conv3": None}
    assert flags((128, 128)) == {"conv1": None, "conv2": None, "conv3": False}      # grids 64, 32, 16
    assert flags((32, 32)) == {"conv1": False, "conv2": False, "conv3": False}      # the small parity shapes
    assert flags((256, 96)) == {"conv1": None, "conv2": None, "conv3": False}       # min side decides: 48, 24, 12
    from satlas_super_resolution_amd import hip   # split-bf16 mode: the alternative is the exact fp32 MFMA, 16 rows are enough
    x3 = {s.name: s.s2d for s in engine.discriminator_specs(3, 64, in_hw=(128, 128), dtype=hip.F32X3) if s.k == 4}
    assert x3 == {"conv1": None, "conv2": None, "conv3": None}
    names = [s.name for s in engine.discriminator_specs(3, 64)]
    assert names == ["conv%d" % i for i in range(10)]


def test_every_shipped_option_file_constructs_or_names_what_it_lacks():
    """/root/reference/ssr/options/*.yml (dumped by oracle/make_option_fixtures.py): every file either yields a StepConfig
    or raises NotImplementedError naming the out-of-scope key/type — nothing is silently dropped.  (GPU-free: the model
    constructor runs exactly this function on `opt`.)"""
    import json
    import os
    from conftest import GOLDEN
    from satlas_super_resolution_amd.models.ssr_esrgan_model import _arch_kwargs, step_config_from_opt
    opts = json.load(open(os.path.join(GOLDEN, "ssr_options.json")))
    assert len(opts) == 12
    outcome = {}
    for name, opt in sorted(opts.items()):
        try:
            if opt.get("model_type") != "SSRESRGANModel":
                raise NotImplementedError(f"model_type {opt.get('model_type')}")
            _arch_kwargs(opt["network_g"], "SSR_RRDBNet")
            if "network_d" in opt:
                _arch_kwargs(opt["network_d"], "SSR_UNetDiscriminatorSN")
            cfg = step_config_from_opt(opt)
            outcome[name] = cfg
        except NotImplementedError as e:
            outcome[name] = str(e)
    runs = {k for k, v in outcome.items() if not isinstance(v, str)}
    assert runs == {"allbands_esrgan_s2naip_urban.yml", "esrgan_s2naip_full.yml", "esrgan_s2naip_urban.yml",
                    "old-naip_esrgan_s2naip_urban.yml", "rand_crop_esrgan_s2naip_urban.yml", "infer_example.yml",
                    "infer_grid_example.yml"}, outcome
    assert "clip_opt" in outcome["cliploss_esrgan_s2naip_urban.yml"]
    assert "ssim_opt" in outcome["ssimloss_esrgan_s2naip_urban.yml"]
    for k in ("highresnet_s2naip_urban.yml", "srcnn_s2naip_urban.yml", "osm_obj_esrgan.yml"):
        assert "model_type" in outcome[k]
    cfg = outcome["esrgan_s2naip_urban.yml"]
    assert (cfg.l1_weight, cfg.gan_weight, cfg.lr_g, cfg.lr_d, cfg.betas, cfg.ema_decay) == (1.0, 0.1, 1e-4, 1e-4, (0.9, 0.99), 0.999)
    assert cfg.l1_gt_usm and cfg.percep_gt_usm and not cfg.gan_gt_usm and cfg.feed_disc_lr
    assert cfg.perceptual["type"] == "PerceptualLoss" and cfg.perceptual["layer_weights"]["conv5_4"] == 1
    # options this path cannot honour are rejected by name
    import copy
    for path, val, key in ((("train", "optim_g", "type"), "SGD", "optim_g.type"), (("train", "optim_d", "weight_decay"), 0.1, "weight_decay"),
                           (("train", "scheduler", "type"), "CosineAnnealingRestartLR", "scheduler.type"),
                           (("train", "gan_opt", "gan_type"), "wgan", "gan_opt"), (("train", "pixel_opt", "type"), "MSELoss", "pixel_opt")):
        o = copy.deepcopy(opts["esrgan_s2naip_urban.yml"])
        d = o
        for p in path[:-1]:
            d = d[p]
        d[path[-1]] = val
        with pytest.raises(NotImplementedError, match=key.replace(".", r"\.")):
            step_config_from_opt(o)


def _mini_opt(**over):
    import os
    from conftest import GOLDEN
    mini = os.path.join(GOLDEN, "s2naip_mini")
    opt = {"phase": "train", "scale": 4, "name": "mini", "type": "S2NAIPDataset", "sentinel2_path": os.path.join(mini, "sentinel2"),
           "naip_path": os.path.join(mini, "naip")}
    opt.update(over)
    return opt, mini


def test_s2naip_dataset_matches_reference_samples():
    """satlas_super_resolution_amd.data.S2NAIPDataset on the committed miniature dataset against the samples the UNMODIFIED
    reference class returned for it (oracle/make_dataset_golden.py -> tests/golden/s2naip_samples.pt): same frames picked under the
    same `random` seed, black-pixel rejection, rand_crop, use_3d, extra bands, old_hr; and the skip rules on the invalid chips."""
    import os
    import random
    from satlas_super_resolution_amd.registry import build_dataset
    import satlas_super_resolution_amd.data  # noqa: F401  (registers the dataset)
    fx = load_golden("s2naip_samples")
    configs = {"plain": dict(n_s2_images=8), "rand_crop": dict(n_s2_images=8, rand_crop=True), "use_3d": dict(n_s2_images=4, use_3d=True),
               "bands": dict(n_s2_images=8, s2_bands=["b08", "tci"]), "old_hr": dict(n_s2_images=8, old_naip_path="OLD")}
    seen = 0
    for cname, over in configs.items():
        opt, mini = _mini_opt(**over)
        if over.get("old_naip_path"):
            opt["old_naip_path"] = os.path.join(mini, "old_naip")
        ds = build_dataset(opt)
        assert len(ds) == 5
        by_chip = {dp[2]: i for i, dp in enumerate(ds.datapoints)}
        for (c, chip), ref in fx.items():
            if c != cname:
                continue
            random.seed(ref["seed"])
            s = ds[by_chip[chip]]
            assert s["Chip"] == chip and s["Phase"] == "train" and s["Index"] == by_chip[chip]
            for k in ("hr", "lr", "old_hr"):
                if k in ref:
                    assert s[k].dtype == torch.uint8 and s[k].shape == ref[k].shape and torch.equal(s[k], ref[k]), (cname, chip, k)
            assert ("old_hr" in s) == ("old_hr" in ref)
            seen += 1
        # invalid datapoints are skipped the way the reference does: index += number of skips so far, wrapping to 0
        valid = lambda i: ds.datapoints[i][2] not in (("101_200", "102_200") if over["n_s2_images"] == 8 else ("101_200",)) \
            and not (cname == "bands" and ds.datapoints[i][2] == "101_201")
        for bad in ("101_200",):
            i, k = by_chip[bad], 0
            while not valid(i):
                k += 1
                i = i + k
                if i >= len(ds):
                    i = 0
            random.seed(5)
            assert ds[by_chip[bad]]["Index"] == i
    assert seen == len(fx) == 14
    w = ds.get_tile_weight_sampler({"100_200": 5.0})
    assert sorted(w.weights.tolist()) == [1.0, 1.0, 1.0, 1.0, 5.0] and len(list(iter(w))) == 5


def test_infer_grid_driver_io_and_sharding(tmp_path):
    """satlas_super_resolution_amd.infer_grid.run_infer_grid (the reference's infer_grid.py flow: PNG chunks in, 128x128 PNG chunks
    + stitched tiles out) with a stand-in model on the CPU: two ranks write disjoint chunk files that together equal the
    one-rank run, file names keep tile / index, a complete 16x16 tile is stitched (2048x2048 SR, 512x512 Sentinel-2), an incomplete
    one is skipped."""
    import numpy as np
    from PIL import Image
    from satlas_super_resolution_amd.infer_grid import run_infer_grid
    rng = np.random.RandomState(0)
    data = tmp_path / "sentinel2"
    for tile, n in (("7_9", 256), ("8_9", 5)):
        (data / tile).mkdir(parents=True)
        for k in range(n):
            i, j = divmod(k, 16)
            Image.fromarray(rng.randint(1, 256, (10 * 32, 32, 3)).astype(np.uint8)).save(data / tile / f"{i}_{j}.png")
    model = lambda x: x[:, :3].repeat_interleave(4, 2).repeat_interleave(4, 3)       # first picked frame, nearest x4
    import random
    outs = {}
    for tag, world in (("one", 1), ("two", 2)):
        save = tmp_path / f"out_{tag}"
        opt = {"data_dir": str(data), "n_lr_images": 8, "save_path": str(save), "batch": 37}
        for rank in reversed(range(world)):       # sequential stand-in for the barrier: rank 0 (which stitches) runs last
            random.seed(3)
            res = run_infer_grid(opt, model=model, rank=rank, world=world, device=torch.device("cpu"))
            if rank == 1:
                assert (res["chunks"], res["tiles_stitched"]) == (130, 0)
        outs[tag] = save
        assert (save / "7_9" / "stitched_sr.png").exists() and (save / "7_9" / "stitched_s2.png").exists()
        assert not (save / "8_9" / "stitched_sr.png").exists() and len(list((save / "8_9").glob("*.png"))) == 5
        sr = np.asarray(Image.open(save / "7_9" / "stitched_sr.png"))
        s2 = np.asarray(Image.open(save / "7_9" / "stitched_s2.png"))
        assert sr.shape == (2048, 2048, 3) and s2.shape == (512, 512, 3)
        first = np.asarray(Image.open(data / "7_9" / "3_5.png")).reshape(-1, 32, 32, 3)[0]
        assert (s2[96:128, 160:192] == first).all()                                   # cell (3, 5) of the Sentinel-2 mosaic
        assert (sr[384:512, 640:768] == np.asarray(Image.open(save / "7_9" / "3_5.png"))).all()
    assert (res["chunks"], res["tiles_stitched"]) == (131, 1)
    assert len(list((outs["two"] / "7_9").glob("[0-9]*_[0-9]*.png"))) == 256


def test_wgrad_items_are_paired_without_losing_or_duplicating_work():
    """Host logic of the batched weight-gradient launch (engine.WgradBatch): blocks of 32 output-gradient channels that are
    contracted with the same input patch share one work item (two dY planes per X patch, csrc/wgrad_bf16.hip).  The pairing
    must keep every (layer, co block, ci chunk, pixel range) exactly once, pair only items that read the same x view, ci chunk,
    geometry and number of valid input channels, and the launch order must start with the longest items."""
    from satlas_super_resolution_amd import engine, hip
    V = hip.View
    wb = engine.WgradBatch(hip.BF16, 3, 1)
    base = 0x10000000
    # two dense blocks (conv1..conv5 read one 192-channel buffer: rrdbnet_arch.py:37-42) + one 64 -> 64 layer at 128 x 128
    for r in range(2):
        xb = base + r * 0x4000000
        for k in range(5):
            cin, cout = 64 + 32 * k, (64 if k == 4 else 32)
            dy = V(xb + 0x1000000, 192, 64 + 32 * k) if k < 4 else V(xb + 0x2000000, 192, 0)
            wb.add(V(xb, 192, 0), dy, 4, 32, 32, 1, cin, cout, 32, 32, 1.0, 0x5000 + 64 * k, cin, 0x6000)
    wb.add(V(base + 0x9000000, 64, 0), V(base + 0xa000000, 64, 0), 4, 128, 128, 1, 64, 64, 128, 128, 1.0, 0x7000, 64, 0x8000)
    singles = [(it.layer, it.co0, it.ci0, it.tile_begin, it.tile_end) for it in wb.items]
    assert len(singles) == len(set(singles)) == 2 * 14 + 2 * (4 * 64 // 128)        # 256 tiles of 16 x 16 in two ranges, two co blocks
    paired = wb._pair(wb.items)
    seen = []
    for it in paired:
        seen.append((it.layer, it.co0, it.ci0, it.tile_begin, it.tile_end))
        if it.nco == 2:
            seen.append((it.layer_b, it.co0_b, it.ci0, it.tile_begin, it.tile_end))
            A, B = wb.layers[it.layer], wb.layers[it.layer_b]
            assert (A.x.p, A.x.cs, A.x.coff) == (B.x.p, B.x.cs, B.x.coff)
            assert (A.N, A.Hi, A.Wi, A.up, A.Gh, A.Gw) == (B.N, B.Hi, B.Wi, B.up, B.Gh, B.Gw)
            assert (min(64, A.Cin_w - it.ci0) > 32) == (min(64, B.Cin_w - it.ci0) > 32)
            assert (it.layer, it.co0) != (it.layer_b, it.co0_b)
    assert sorted(seen) == sorted(singles)                       # nothing lost, nothing twice
    # a dense block: 6 pairs + the two 32-channel leftovers (dpre2 over x1, dpre4 over x3); the 64 -> 64 layer: its two halves
    per_block = [it for it in paired if it.layer < 5]
    assert sum(it.nco == 2 for it in per_block) == 6 and sum(it.nco != 2 for it in per_block) == 2
    assert all(it.nco == 2 and it.layer_b == it.layer and {it.co0, it.co0_b} == {0, 32} for it in paired if it.layer == 10)
    # launch order: longest first (workgroups are handed out in index order)
    order = sorted(paired, key=lambda it: -wb._cost(it))
    costs = [wb._cost(it) for it in order]
    assert costs == sorted(costs, reverse=True) and order[0].nco == 2
    # the simulated balance never makes the launch longer
    assert wb._makespan(wb._balance(paired)) <= wb._makespan(paired)


def test_wgrad_batches_of_the_split_mode_pick_the_one_pass_kernels(monkeypatch):
    """engine.WgradBatch in the fp32x3 mode: 3x3 stride-1 layers read the fp32 buffers themselves (csrc/wgrad_x3.hip: 8 x 16-pixel
    tiles, paired 64-channel items, as many pixels per item as the bf16 kernel's), and since round 6 the 4x4 stride-2 layers too (4 x 16-pixel
    tiles, 32 x 32-channel items); SSR_X3_WGRAD_FUSED=0 / SSR_X3_WGRAD_FUSED4=0 restore the split pass + three bf16 launches; the pairing rules
    are the bf16 kernel's."""
    from satlas_super_resolution_amd import engine, hip
    V = hip.View
    monkeypatch.delenv("SSR_X3_WGRAD_FUSED", raising=False)
    monkeypatch.delenv("SSR_X3_WGRAD_FUSED4", raising=False)
    wb = engine.WgradBatch(hip.F32X3, 3, 1)
    assert wb.kdt == hip.F32X3
    assert engine.WgradBatch(hip.F32X3, 4, 2).kdt == hip.F32X3 and engine.WgradBatch(hip.BF16, 3, 1).kdt == hip.BF16
    assert engine.WgradBatch(hip.F32, 3, 1).kdt == hip.F32
    monkeypatch.setenv("SSR_X3_WGRAD_FUSED", "0")
    monkeypatch.setenv("SSR_X3_WGRAD_FUSED4", "0")
    assert engine.WgradBatch(hip.F32X3, 3, 1).kdt == hip.BF16 and engine.WgradBatch(hip.F32X3, 4, 2).kdt == hip.BF16
    monkeypatch.delenv("SSR_X3_WGRAD_FUSED")
    monkeypatch.delenv("SSR_X3_WGRAD_FUSED4")
    w4 = engine.WgradBatch(hip.F32X3, 4, 2)                       # D conv2 at B = 32: 128 -> 256 channels, 64 x 64 -> 32 x 32
    w4.add(V(0x10000000, 128, 0), V(0x20000000, 256, 0), 32, 64, 64, 1, 128, 256, 32, 32, 1.0, 0x9000, 128, None)
    assert hip.lib().ssr_wgrad_tiles(32, 32, 32, w4.kdt, 4) == 32 * 8 * 2 and hip.lib().ssr_wgrad_tiles(32, 32, 32, hip.BF16, 4) == 32 * 4 * 2
    assert hip.lib().ssr_wgrad_ci_tile(w4.kdt, 4) == 32 and len({(it.co0, it.ci0) for it in w4.items}) == 8 * 4
    assert all(it.tile_end - it.tile_begin == 128 for it in w4.items)            # 128 tiles of 4 x 16 pixels: the bf16 kernel's 64 of 8 x 16
    base = 0x10000000
    for k in range(5):                                            # one dense block at B = 16 (a half-batch chain of the step)
        cin, cout = 64 + 32 * k, (64 if k == 4 else 32)
        dy = V(base + 0x1000000, 192, 64 + 32 * k) if k < 4 else V(base + 0x2000000, 192, 0)
        wb.add(V(base, 192, 0), dy, 16, 32, 32, 1, cin, cout, 32, 32, 1.0, 0x5000 + 64 * k, cin, 0x6000)
    wb.add(V(base + 0x9000000, 64, 0), V(base + 0xa000000, 64, 0), 16, 128, 128, 1, 64, 64, 128, 128, 1.0, 0x7000, 64, 0x8000)
    body = [it for it in wb.items if it.layer < 5]
    assert all((it.tile_begin, it.tile_end) == (0, 16 * 4 * 2) for it in body)        # 8 x 16 tiles: 128 per layer, one pixel range
    assert len(body) == 1 + 2 + 2 + 3 + 2 * 3                                          # ci chunks of 64: 1, 2, 2, 3 and 3 x two co blocks
    big = [it for it in wb.items if it.layer == 5]
    assert len(big) == 2 * (16 * 16 * 8 // 256) and all(it.atomic == 1 for it in big)  # 2048 tiles in ranges of 256, two co blocks
    paired = wb._pair(wb.items)
    assert sum(it.nco == 2 for it in paired if it.layer < 5) == 6                      # as the bf16 kernel: 6 pairs + 2 singles per block
    assert hip.lib().ssr_wgrad_tiles(16, 32, 32, wb.kdt, 3) == 128


def test_arithmetic_mode_codes_split_into_storage_and_forward_codes():
    """compute_dtype names -> mode codes; a mode is (what the C ABI sees for tensors / backward launches, what its forward convolutions carry):
    fp32f = exact forward on split-bf16 storage, fp32h = fp16-split forward (include/ssr_hip.h SSR_F32H) on split-bf16 storage"""
    from satlas_super_resolution_amd import hip
    table = {"fp32": (hip.F32, hip.F32), "bf16": (hip.BF16, hip.BF16), "fp32x3": (hip.F32X3, hip.F32X3), "fp32f": (hip.F32X3, hip.F32),
             "fp32h": (hip.F32X3, hip.F32H3)}
    for name, (st, fw) in table.items():
        m = hip.dtype_code(name)
        assert hip.dtype_code(m) == m and (hip.storage_code(m), hip.forward_code(m)) == (st, fw), name
    assert hip.F32H3 == 3 and len({hip.dtype_code(n) for n in table}) == 5
    import re
    hdr = open(__file__.replace("tests/test_host_boundary.py", "include/ssr_hip.h")).read()
    assert re.search(r"#define SSR_F32H 3\b", hdr) and re.search(r"#define SSR_F32H_WSHIFT 10\b", hdr)
    with pytest.raises(ValueError):
        hip.dtype_code("fp16")
    # the stride-2 layers of the discriminator take the space-to-depth form from 16-row grids on in both split forward arithmetics
    from satlas_super_resolution_amd import engine
    for dt in (hip.F32X3, hip.F32H3, hip.dtype_code("fp32h")):
        assert {s.name: s.s2d for s in engine.discriminator_specs(3, 64, in_hw=(128, 128), dtype=dt) if s.k == 4}["conv3"] is None
    assert {s.name: s.s2d for s in engine.discriminator_specs(3, 64, in_hw=(128, 128), dtype=hip.F32) if s.k == 4}["conv3"] is False


def test_wgrad_hybrid_item_order_is_a_cost_sorted_permutation_with_buffer_groups_on_one_xcd(monkeypatch):
    """engine.WgradBatch, SSR_WGRAD_ORDER=hybrid: the items of a launch stay sorted by cost (list scheduling), and inside a run of equal cost the items
    that read one buffer over the same tiles sit on ONE XCD (block index % 8), next to each other in its queue"""
    from satlas_super_resolution_amd import engine, hip
    V = hip.View
    monkeypatch.setattr(engine, "_device_cus", lambda: (256, 8))
    monkeypatch.setenv("SSR_WGRAD_ORDER", "hybrid")
    wb = engine.WgradBatch(hip.F32X3, 3, 1)
    for blk in range(24):                                         # 24 dense blocks at B = 16, each over its own 192-channel buffer
        base = 0x10000000 + blk * 0x4000000
        for k in range(5):
            cin, cout = 64 + 32 * k, (64 if k == 4 else 32)
            dy = V(base + 0x1000000, 192, 64 + 32 * k) if k < 4 else V(base + 0x2000000, 192, 0)
            wb.add(V(base, 192, 0), dy, 16, 32, 32, 1, cin, cout, 32, 32, 1.0, 0x5000 + 64 * (5 * blk + k), cin, 0x6000)
    ref = wb._pair(list(wb.items))
    key = lambda it: (it.layer, it.co0, it.ci0, it.tile_begin, it.tile_end, it.nco, it.layer_b, it.co0_b)
    got = wb._cost_xcd_order(ref)
    assert sorted(map(key, got)) == sorted(map(key, ref))                                   # a permutation
    costs = [wb._cost(it) for it in got]
    assert costs == sorted(costs, reverse=True)                                             # longest first
    same_xcd = 0
    groups = {}
    for pos, it in enumerate(got):
        groups.setdefault((wb._cost(it), wb.layers[it.layer].x.p), []).append(pos % 8)
    for xs in groups.values():
        same_xcd += len(set(xs)) == 1
    assert same_xcd >= 0.9 * len(groups), (same_xcd, len(groups))                           # (the last groups of a run fill the short queues)
