"""GPU: the drop-in boundary behaves like the reference loop expects (train.py:62-133), beyond the plain step:
gated iterations (net_d_iters / net_d_init_iters) incl. EMA, `old_hr` discriminator input, BasicSR-shaped checkpoint
files and resume before the first batch, learning-rate schedule across step rebuilds."""
import copy
from collections import OrderedDict

import pytest
import torch

from conftest import load_golden, parity_close, rel_err

pytestmark = pytest.mark.gpu


def _tiny(c_in=6, c_d=3, nb=1):
    from oracle import esrgan_oracle as O
    g_kw = dict(num_in_ch=c_in, num_out_ch=3, scale=4, num_feat=16, num_block=nb, num_grow_ch=8)
    d_kw = dict(num_in_ch=c_d, num_feat=8, skip_connection=True)
    return g_kw, d_kw, O.generator_init(seed=31, **g_kw), O.discriminator_init(c_d, 8, seed=32)


def _close_update(got, ref, p0, what, frac=2e-2, lr_steps=None):
    """Post-step parameters compared on the UPDATE.  With `lr_steps` (= lr x number of Adam steps): Adam normalises the gradient,
    so an element whose gradient is at rounding level gets a +-lr step whose size is noise in any fp32 implementation - at most
    0.1 % of a tensor's elements may leave the tight bound, and none by more than the maximal Adam step (as
    tests/test_gpu_parity.py::_update_close)."""
    upd_ref, upd = ref - p0, got - p0
    err = (upd - upd_ref).abs()
    tight = frac * upd_ref.abs().max() + 3e-7 * ref.abs().max() + 1e-9
    if lr_steps is None:
        assert err.max() <= tight, what
        return
    assert (err > tight).float().mean().item() <= 1e-3, (what, "fraction of elements off", (err > tight).float().mean().item())
    assert err.max() <= 2.1 * lr_steps + tight, (what, float(err.max()))


def test_gated_iterations_follow_the_reference_incl_ema():
    """net_d_iters=2, net_d_init_iters=1 (ssr_esrgan_model.py:144): G steps only on iterations 2 and 4 of 4, D on all of
    them, the D forward of the G phase (and its power iteration) is skipped on gated iterations, model_ema runs on EVERY
    iteration (:230-231).  Eager and hipGraph replay."""
    from oracle import esrgan_oracle as O
    from satlas_super_resolution_amd.train_step import ESRGANTrainStep, StepConfig
    g_kw, d_kw, g0, d0 = _tiny()
    torch.manual_seed(5)
    data = [(torch.rand(2, 6, 8, 8), torch.rand(2, 3, 32, 32)) for _ in range(4)]
    for use_graph in (False, True):
        orc = O.ESRGANOracle(g0, d0, O.StepConfig(net_d_iters=2, net_d_init_iters=1, ema_decay=0.9, lr_g=1e-3, lr_d=1e-3))
        ts = ESRGANTrainStep(g_kw, d_kw, 2, 8, 8, "fp32",
                             StepConfig(net_d_iters=2, net_d_init_iters=1, ema_decay=0.9, lr_g=1e-3, lr_d=1e-3), use_graph=use_graph)
        ts.load_state(g0, d0)
        # each graph is captured on its second use: run the schedule twice so that gated and full iterations both replay
        for rep in range(2):
            for it, (lr, gt) in enumerate(data, start=1 + 4 * rep):
                cur = it if rep == 0 else it       # iterations 5..8: 6 and 8 are G iterations
                ref_log = orc.step(lr, gt, cur)
                ts.feed_data(lr.cuda(), gt.cuda())
                ts.step(cur)
                log = ts.log()
                g_on = cur % 2 == 0 and cur > 1
                assert ("l_g_pix" in ref_log) == g_on
                for k, v in ref_log.items():
                    assert abs(log[k] - v) <= 2e-3 * max(1.0, abs(v)), (use_graph, cur, k, log[k], v)
                if not g_on:
                    assert log["l_g_pix"] == 0.0 and log["l_g_gan"] == 0.0
        ema = ts.ema_state_dict()
        for k in g0:
            _close_update(ts.g_store.tensor(k).cpu(), orc.g[k], g0[k], ("G", k), 5e-2)
            _close_update(ema[k].cpu(), orc.g_ema[k], g0[k], ("EMA", k), 5e-2)
        # 8 iterations: the EMA moved although G stepped on only 4 of them; u/v advanced 2 (D phases) + 1 (G phase) per full iteration
        for n in O.SN_LAYERS:
            assert rel_err(ts.d_store.u[n], orc.d[n + ".weight_u"]) < 2e-3, n


@pytest.mark.parametrize("feed_disc_lr", [False, True])
def test_old_hr_discriminator_input(feed_disc_lr):
    """`old_hr` in the batch: D sees [img | lr_resized | old_hr] (ssr_esrgan_model.py:171-175, 202-207)."""
    from oracle import esrgan_oracle as O
    from satlas_super_resolution_amd.train_step import ESRGANTrainStep, StepConfig
    c_d = 3 + (6 if feed_disc_lr else 0) + 3
    g_kw, d_kw, g0, d0 = _tiny(6, c_d)
    torch.manual_seed(6)
    lr, gt, old = torch.rand(2, 6, 8, 8), torch.rand(2, 3, 32, 32), torch.rand(2, 3, 32, 32)
    orc = O.ESRGANOracle(g0, d0, O.StepConfig(feed_disc_lr=feed_disc_lr))
    ref_log = orc.step(lr, gt, 1, old_hr=old)
    ts = ESRGANTrainStep(g_kw, d_kw, 2, 8, 8, "fp32", StepConfig(feed_disc_lr=feed_disc_lr, old_hr=True), use_graph=False)
    ts.load_state(g0, d0)
    ts.feed_data(lr.cuda(), gt.cuda(), old_hr=old.cuda())
    ts.step(1)
    log = ts.log()
    for k, v in ref_log.items():
        assert abs(log[k] - v) <= 1e-3 * max(1.0, abs(v)), (k, log[k], v)
    for k, g in orc.g_grads.items():
        assert parity_close(ts.g_store.tensor(k, ts.g_store.grad), g), ("g grad", k)
    for k, g in orc.d_grads.items():
        assert parity_close(ts.d_store.tensor(k, ts.d_store.grad), g), ("d grad", k)
    with pytest.raises(AssertionError):
        ts.feed_data(lr.cuda(), gt.cuda())            # a batch without old_hr cannot feed a D built for it


def _opt(tmp_path, fx, **over):
    opt = {
        "model_type": "SSRESRGANModel", "scale": 4, "manual_seed": 0, "is_train": True, "dist": False, "name": "t",
        # these tests pin the plugin's CONTROL FLOW to fixtures of the reference's own methods through Adam-normalised parameter updates
        # (a relative error of a near-zero gradient is amplified to an update of +-lr): exact arithmetic, not the plugin's default fp32h
        "compute_dtype": "fp32",
        "l1_gt_usm": False, "percep_gt_usm": False, "gan_gt_usm": False, "feed_disc_lr": False,
        "network_g": dict(type="SSR_RRDBNet", **fx["g_kwargs"]),
        "network_d": dict(type="SSR_UNetDiscriminatorSN", **fx["d_kwargs"]),
        "path": {"models": str(tmp_path / "models"), "training_states": str(tmp_path / "states"),
                 "visualization": str(tmp_path / "vis")},
        "train": {"ema_decay": 0.999, "optim_g": {"type": "Adam", "lr": 1e-4, "weight_decay": 0, "betas": [0.9, 0.99]},
                  "optim_d": {"type": "Adam", "lr": 2e-4, "weight_decay": 0, "betas": [0.5, 0.9]},
                  "scheduler": {"type": "MultiStepLR", "milestones": [3], "gamma": 0.5},
                  "pixel_opt": {"type": "L1Loss", "loss_weight": 1.0, "reduction": "mean"},
                  "gan_opt": {"type": "GANLoss", "gan_type": "vanilla", "real_label_val": 1.0, "fake_label_val": 0.0, "loss_weight": 0.1},
                  "net_d_iters": 1, "net_d_init_iters": 0},
        "val": {"val_freq": 2, "save_img": True,
                "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 4, "test_y_channel": False},
                            "ssim": {"type": "calculate_ssim", "crop_border": 4, "test_y_channel": False},
                            "cpsnr": {"type": "calculate_cpsnr", "crop_border": 4, "test_y_channel": False}}},
    }
    opt.update(over)
    return opt


def _batch(lr, gt):
    return {"lr": (lr * 255).round().to(torch.uint8), "hr": (gt * 255).round().to(torch.uint8)}


def test_save_resume_round_trip_before_first_batch(tmp_path):
    """train.py:62-65: build_model(opt) then model.resume_training(state) BEFORE any feed_data.  A run that is saved after
    2 iterations, rebuilt from its files and resumed must continue exactly like the uninterrupted run (parameters, Adam
    moments, step count, EMA from 'params_ema', learning-rate schedule); the .state file has BasicSR's shape."""
    from satlas_super_resolution_amd import models  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    fx = load_golden("step_tiny")
    data = list(fx["data"]) * 2          # 4 iterations
    opt = _opt(tmp_path, fx)
    a = build_model(opt)
    for it in (1, 2):
        a.update_learning_rate(it)
        a.feed_data(_batch(*data[it - 1]))
        a.optimize_parameters(it)
    a.save(0, 2)
    state = torch.load(tmp_path / "states" / "2.state", weights_only=False)
    assert set(state) == {"epoch", "iter", "optimizers", "schedulers"} and state["iter"] == 2
    assert len(state["optimizers"]) == 2 and len(state["schedulers"]) == 2
    og = state["optimizers"][0]
    assert set(og) == {"state", "param_groups"} and og["param_groups"][0]["betas"] == (0.9, 0.99)
    assert state["optimizers"][1]["param_groups"][0]["betas"] == (0.5, 0.9)          # optim_d's own betas
    n_g = len(a.ts.g_store.offsets)
    assert og["param_groups"][0]["params"] == list(range(n_g)) and set(og["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(og["state"][n_g - 1]["step"]) == 2.0
    assert og["state"][0]["exp_avg"].shape == a.ts.g_store.tensor("conv_first.weight").shape
    assert state["schedulers"][0]["last_epoch"] == 1 and state["schedulers"][0]["milestones"] == {3: 1}
    # torch's own Adam accepts the optimizer state (that is what BasicSR's resume_training does)
    plist = [torch.nn.Parameter(a.ts.g_store.tensor(k).cpu().clone()) for k in a.ts.g_store.offsets]
    torch.optim.Adam(plist, lr=1e-4, betas=(0.9, 0.99)).load_state_dict(og)
    for it in (3, 4):
        a.update_learning_rate(it)
        a.feed_data(_batch(*data[it - 1]))
        a.optimize_parameters(it)
    # ---- second process: rebuild from the files, resume before the first batch
    # through train.py's resolve_resume, with the key the shipped option files carry (`param_key_g: params_ema`,
    # esrgan_s2naip_urban.yml:86-92): the resume must load the TRAINED weights ('params') into the generator - the Adam moments
    # belong to them - and the EMA from 'params_ema' (BasicSR check_resume resets the key)
    from satlas_super_resolution_amd.train import resolve_resume
    opt_b = copy.deepcopy(opt)
    opt_b["path"].update(pretrain_network_g="published/esrgan.pth", param_key_g="params_ema", strict_load_g=True,
                         resume_state=str(tmp_path / "states" / "2.state"))
    state_b = resolve_resume(opt_b, log=lambda m: None)
    assert state_b["iter"] == 2 and opt_b["path"]["param_key_g"] == "params"
    assert opt_b["path"]["pretrain_network_g"] == str(tmp_path / "models" / "net_g_2.pth")
    assert opt_b["path"]["pretrain_network_d"] == str(tmp_path / "models" / "net_d_2.pth")
    ck = torch.load(tmp_path / "models" / "net_g_2.pth", map_location="cpu")
    assert (ck["params"]["conv_first.weight"] - ck["params_ema"]["conv_first.weight"]).abs().max() > 0
    b = build_model(opt_b)
    b.resume_training(state)
    assert b.ts is None
    for it in (3, 4):
        b.update_learning_rate(it)
        b.feed_data(_batch(*data[it - 1]))
        b.optimize_parameters(it)
        la, lb = a.get_current_log(), b.get_current_log()
    assert b.get_current_learning_rate() == a.get_current_learning_rate() == [0.5e-4]     # milestone 3 reached at iteration 4
    assert float(b.ts.opt_d.lr.item()) == pytest.approx(1e-4)
    assert int(b.ts.opt_g.step.item()) == 4
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(la[k])), (k, la[k], lb[k])
    for sa, sb, what in ((a.ts.g_store.data, b.ts.g_store.data, "G"), (a.ts.d_store.data, b.ts.d_store.data, "D"),
                         (a.ts.opt_g.ema, b.ts.opt_g.ema, "EMA"), (a.ts.opt_g.exp_avg, b.ts.opt_g.exp_avg, "m"),
                         (a.ts.opt_d.exp_avg_sq, b.ts.opt_d.exp_avg_sq, "v")):
        assert rel_err(sb, sa) < 1e-4, what
    assert (a.ts.opt_g.ema - a.ts.g_store.data).abs().max() > 0        # the EMA history really differs from the weights


def test_learning_rate_survives_a_step_rebuild(tmp_path):
    """update_learning_rate() precedes feed_data() in the loop (train.py:106-108): the device LR must be the scheduled one on
    the very first iteration (warm-up) and after a rebuild for another batch shape (ragged last batch)."""
    from satlas_super_resolution_amd import models  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    fx = load_golden("step_tiny")
    lr, gt = fx["data"][0]
    m = build_model(_opt(tmp_path, fx))
    m.update_learning_rate(1, warmup_iter=10)
    m.feed_data(_batch(lr, gt))
    assert float(m.ts.opt_g.lr.item()) == pytest.approx(1e-5) and float(m.ts.opt_d.lr.item()) == pytest.approx(2e-5)
    m.optimize_parameters(1)
    m.update_learning_rate(5, warmup_iter=-1)            # past milestone 3
    m.feed_data(_batch(lr[:1], gt[:1]))                  # batch of 1: the step is rebuilt
    assert m.ts.B == 1 and float(m.ts.opt_g.lr.item()) == pytest.approx(0.5e-4)
    assert int(m.ts.opt_g.step.item()) == 1              # optimizer state carried over
    m.optimize_parameters(5)
    assert all(v == v for v in m.get_current_log().values())


def test_validation_runs_metrics_and_writes_images(tmp_path):
    """nondist_validation (ssr_esrgan_model.py:269-352) with psnr / ssim / cpsnr on device and PNG output."""
    import numpy as np
    from PIL import Image
    from oracle import metrics_oracle as MO
    from satlas_super_resolution_amd import models  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    fx = load_golden("step_tiny")
    lr, gt = fx["data"][0]
    m = build_model(_opt(tmp_path, fx))
    m.update_learning_rate(1)
    m.feed_data(_batch(lr, gt))
    m.ts.load_state(fx["g0"], fx["d0"])
    m.optimize_parameters(1)

    class DS:
        opt = {"name": "val"}

    class Loader(list):
        dataset = DS()

    loader = Loader([_batch(lr[:1], gt[:1]), _batch(lr[1:], gt[1:])])
    res = m.validation(loader, 1, None, save_img=True) or m.metric_results
    assert set(res) == {"psnr", "ssim", "cpsnr"}
    # reference values from the CPU restatement on the images the model wrote
    vals = {"psnr": [], "ssim": [], "cpsnr": []}
    for idx in range(2):
        sr = np.asarray(Image.open(tmp_path / "vis" / str(idx) / f"{idx}_1.png"))
        g = np.asarray(Image.open(tmp_path / "vis" / str(idx) / f"{idx}_1_gt.png"))
        assert sr.shape == (32, 32, 3) and g.shape == (32, 32, 3)
        assert (g == (gt[idx] * 255).round().permute(1, 2, 0).numpy().astype(np.uint8)).all()
        vals["psnr"].append(MO.calculate_psnr(sr, g, 4))
        vals["ssim"].append(MO.calculate_ssim(sr, g, 4))
        vals["cpsnr"].append(MO.calculate_cpsnr(sr, g, 4))
    for k in vals:
        assert res[k] == pytest.approx(sum(vals[k]) / 2, rel=1e-6, abs=1e-6), k
    assert m.best_metric_results["val"]["psnr"]["iter"] == 1


def test_device_metrics_match_the_cpu_restatement():
    """csrc/metrics.hip through the C ABI: psnr / ssim / cpsnr on uint8 images vs oracle/metrics_oracle.py (cpsnr additionally
    vs the values the unmodified reference function produced, tests/golden/cpsnr.pt); ragged sizes, crop 0 and 4."""
    import numpy as np
    from oracle import metrics_oracle as MO
    from satlas_super_resolution_amd import metrics as M
    cases = [(c["img"].numpy(), c["img2"].numpy(), c["crop_border"], c["value"]) for c in load_golden("cpsnr")]
    rng = np.random.RandomState(3)
    for (h, w, crop) in [(128, 128, 4), (37, 53, 0), (32, 32, 4)]:
        a = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        b = np.clip(a.astype(np.int32) + rng.randint(-9, 10, a.shape), 0, 255).astype(np.uint8)
        cases.append((a, b, crop, None))
    for a, b, crop, ref_cpsnr in cases:
        ta, tb = torch.from_numpy(a).cuda()[None], torch.from_numpy(b).cuda()[None]
        assert M.calculate_psnr(ta, tb, crop) == pytest.approx(MO.calculate_psnr(a, b, crop), rel=1e-12)
        assert M.calculate_ssim(ta, tb, crop) == pytest.approx(MO.calculate_ssim(a, b, crop), rel=1e-9)
        v = M.calculate_cpsnr(ta, tb, crop)
        assert v == pytest.approx(MO.calculate_cpsnr(a, b, crop), rel=1e-10)
        if ref_cpsnr is not None:
            assert v == pytest.approx(ref_cpsnr, rel=1e-10)
    same = torch.from_numpy(cases[0][0]).cuda()[None]
    assert M.calculate_psnr(same, same, 4) == float("inf") and M.calculate_cpsnr(same, same, 4) == float("inf")
    with pytest.raises(NotImplementedError):
        M.calculate_psnr(same, same, 4, test_y_channel=True)


def test_quantize_u8_round_and_truncate_bit_exact():
    """tensor2img's round-half-even and infer_grid.py's truncating astype(uint8), bit for bit against numpy."""
    import numpy as np
    from satlas_super_resolution_amd import metrics as M
    torch.manual_seed(0)
    x = torch.rand(2, 3, 17, 23) * 1.2 - 0.1
    x[0, 0, 0, :8] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255, 1.0, 0.0, -3.0, 7.0])
    xn = np.clip(x.numpy(), 0, 1)
    ref_round = (xn * np.float32(255.0)).round().astype(np.uint8).transpose(0, 2, 3, 1)
    ref_trunc = (xn * np.float32(255)).astype(np.uint8).transpose(0, 2, 3, 1)
    assert (M.tensor2img_u8(x.cuda()).cpu().numpy() == ref_round).all()
    assert (M.tensor2img_u8(x.cuda(), truncate=True).cpu().numpy() == ref_trunc).all()


@pytest.mark.parametrize("compute_dtype", ["fp32x3", "bf16", "fp32h", pytest.param("fp32f", marks=pytest.mark.slow)])
def test_shipped_option_file_builds_and_trains(tmp_path, monkeypatch, compute_dtype):
    """/root/reference/ssr/options/esrgan_s2naip_urban.yml as shipped (tests/golden/ssr_options.json) driven like train.py does:
    full-size networks (nf=64, nb=23; 36-channel generator input as the file says), L1 + VGG19 perceptual + GAN, USM targets,
    feed_disc_lr, MultiStepLR, EMA.  Only what options.parse_options adds at run time is added (is_train, dist, paths), plus
    network_d.num_in_ch = 3 + 36: the file as shipped pairs `feed_disc_lr: True` with a 3-channel discriminator, which the
    reference cannot run either (torch.cat of 39 channels into conv0 of 3).  VGG19 weights come from a file in
    torchvision's layout (random values here: no network)."""
    import json
    import os
    from conftest import GOLDEN
    from satlas_super_resolution_amd import models, perceptual as P  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    opt = json.load(open(os.path.join(GOLDEN, "ssr_options.json")))["esrgan_s2naip_urban.yml"]
    opt.update(is_train=True, dist=False, rank=0, world_size=1, compute_dtype=compute_dtype)
    opt["path"].update(models=str(tmp_path / "models"), training_states=str(tmp_path / "states"), visualization=str(tmp_path / "vis"))
    with pytest.raises(AssertionError, match="feed_disc_lr"):      # as shipped: the D input width does not match
        bad = build_model(json.loads(json.dumps(opt)))
        bad.feed_data({"lr": torch.zeros(1, 36, 32, 32, dtype=torch.uint8), "hr": torch.zeros(1, 3, 128, 128, dtype=torch.uint8)})
    opt["network_d"]["num_in_ch"] = 3 + opt["network_g"]["num_in_ch"]
    with pytest.raises(FileNotFoundError, match="vgg19"):          # no silent random perceptual network
        m0 = build_model(json.loads(json.dumps(opt)))
        m0.feed_data({"lr": torch.zeros(1, 36, 32, 32, dtype=torch.uint8), "hr": torch.zeros(1, 3, 128, 128, dtype=torch.uint8)})
    wfile = tmp_path / "vgg19-dcbb9e9d.pth"
    torch.save(P.vgg19_random_state(P.vgg19_specs("conv5_4"), seed=1), wfile)
    monkeypatch.setenv("SSR_VGG19_WEIGHTS", str(wfile))
    model = build_model(opt)
    g = torch.Generator().manual_seed(0)
    for it in (1, 2, 3):
        model.update_learning_rate(it, warmup_iter=opt["train"].get("warmup_iter", -1))
        model.feed_data({"lr": torch.randint(0, 256, (2, 36, 32, 32), generator=g, dtype=torch.uint8),
                         "hr": torch.randint(0, 256, (2, 3, 128, 128), generator=g, dtype=torch.uint8)})
        model.optimize_parameters(it)
    log = model.get_current_log()
    assert set(log) == {"l_g_pix", "l_g_percep", "l_g_gan", "l_d_real", "out_d_real", "l_d_fake", "out_d_fake"}
    assert all(v == v and abs(v) < 1e4 for v in log.values()) and log["l_g_percep"] > 0
    assert model.ts.cfg.l1_gt_usm and model.ts.percep_tgt is model.ts.l1_tgt and model.ts.percep_tgt is not model.ts.real_in
    assert model.get_current_learning_rate() == [1e-4]
    model.save(0, 3)
    assert os.path.exists(tmp_path / "models" / "net_g_3.pth") and os.path.exists(tmp_path / "states" / "3.state")
    model.test()
    assert model.get_current_visuals()["result"].shape == (2, 3, 128, 128)


def test_training_loop_on_the_miniature_dataset(tmp_path, monkeypatch):
    """satlas_super_resolution_amd.train.train (the control flow of /root/reference/ssr/train.py:52-140 without BasicSR) end to
    end: S2NAIPDataset (committed miniature set, tile-weight sampler off) -> DataLoader (uint8 batches) -> SSRESRGANModel
    (L1 + VGG19 perceptual + GAN, USM targets, feed_disc_lr) -> logging, checkpoint, validation with psnr / ssim / cpsnr."""
    import os
    from conftest import GOLDEN
    from satlas_super_resolution_amd import perceptual as P
    from satlas_super_resolution_amd.train import train
    mini = os.path.join(GOLDEN, "s2naip_mini")
    ds = {"name": "mini", "type": "S2NAIPDataset", "sentinel2_path": os.path.join(mini, "sentinel2"), "naip_path": os.path.join(mini, "naip"),
          "use_shuffle": False, "num_worker_per_gpu": 0, "batch_size_per_gpu": 2, "n_s2_images": 8}
    wfile = tmp_path / "vgg19.pth"
    torch.save(P.vgg19_random_state(P.vgg19_specs("conv5_4"), seed=1), wfile)
    monkeypatch.setenv("SSR_VGG19_WEIGHTS", str(wfile))
    opt = {
        "name": "mini", "model_type": "SSRESRGANModel", "scale": 4, "manual_seed": 0, "is_train": True, "dist": False,
        "l1_gt_usm": True, "percep_gt_usm": True, "gan_gt_usm": False, "feed_disc_lr": True, "compute_dtype": "bf16",
        "datasets": {"train": dict(ds), "val": dict(ds, name="validation")},
        "network_g": {"type": "SSR_RRDBNet", "num_in_ch": 24, "num_out_ch": 3, "num_feat": 64, "num_block": 2, "num_grow_ch": 32},
        "network_d": {"type": "SSR_UNetDiscriminatorSN", "num_in_ch": 27, "num_feat": 64, "skip_connection": True},
        "path": {"models": str(tmp_path / "models"), "training_states": str(tmp_path / "states"), "visualization": str(tmp_path / "vis")},
        "train": {"ema_decay": 0.999, "optim_g": {"type": "Adam", "lr": 1e-4, "weight_decay": 0, "betas": [0.9, 0.99]},
                  "optim_d": {"type": "Adam", "lr": 1e-4, "weight_decay": 0, "betas": [0.9, 0.99]},
                  "scheduler": {"type": "MultiStepLR", "milestones": [400000], "gamma": 0.5}, "total_iter": 4, "warmup_iter": -1,
                  "pixel_opt": {"type": "L1Loss", "loss_weight": 1.0, "reduction": "mean"},
                  "perceptual_opt": {"type": "PerceptualLoss", "layer_weights": {"conv1_2": 0.1, "conv2_2": 0.1, "conv3_4": 1, "conv4_4": 1, "conv5_4": 1},
                                     "vgg_type": "vgg19", "use_input_norm": True, "perceptual_weight": 1.0, "style_weight": 0,
                                     "range_norm": False, "criterion": "l1"},
                  "gan_opt": {"type": "GANLoss", "gan_type": "vanilla", "real_label_val": 1.0, "fake_label_val": 0.0, "loss_weight": 0.1},
                  "net_d_iters": 1, "net_d_init_iters": 0},
        "val": {"val_freq": 4, "save_img": True, "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 4, "test_y_channel": False},
                                                             "ssim": {"type": "calculate_ssim", "crop_border": 4, "test_y_channel": False},
                                                             "cpsnr": {"type": "calculate_cpsnr", "crop_border": 4, "test_y_channel": False}}},
        "logger": {"print_freq": 2, "save_checkpoint_freq": 4},
    }
    lines = []
    res = train(opt, log=lines.append)
    assert res["iters"] == 4 and set(res["metrics"]) == {"psnr", "ssim", "cpsnr"}
    assert all(v == v for v in res["log"].values()) and "l_g_percep" in res["log"]
    assert 0 < res["metrics"]["psnr"] < 60 and -1 <= res["metrics"]["ssim"] <= 1
    assert any('"iter": 2' in ln for ln in lines) and any('"validation"' in ln for ln in lines)
    for f in ("models/net_g_4.pth", "models/net_d_4.pth", "states/4.state", "models/net_g_latest.pth"):
        assert os.path.exists(tmp_path / f), f
    assert len(os.listdir(tmp_path / "vis")) == 5            # one directory per validation image (5 datapoints in the mini set)


def test_test_pipeline_generator_only(tmp_path):
    """The model plugin built with is_train=False as ssr/test.py:14-46 builds it — generator only, the EMA weights of a
    checkpoint (`param_key_g: params_ema`) — driven over the `test_datasets` of an option file: `model.validation` computes the
    `test.metrics` psnr / cpsnr on the device and writes the images; metrics outside this path raise by name.  The driver is
    satlas_super_resolution_amd/test.py, the BasicSR-free counterpart of the reference's ssr/test.py (as train.py is of ssr/train.py)."""
    import os
    from conftest import GOLDEN
    from oracle import esrgan_oracle as O

    from satlas_super_resolution_amd.test import test_pipeline       # `python -m satlas_super_resolution_amd.test -opt <yml>`
    mini = os.path.join(GOLDEN, "s2naip_mini")
    g_kw = dict(num_in_ch=24, num_out_ch=3, scale=4, num_feat=64, num_block=1, num_grow_ch=32)
    sd = O.generator_init(seed=5, **g_kw)
    torch.save({"params": {k: v + 1.0 for k, v in sd.items()}, "params_ema": sd}, tmp_path / "net_g.pth")
    opt = {"name": "t", "model_type": "SSRESRGANModel", "scale": 4, "manual_seed": 0, "compute_dtype": "fp32x3",
           "test_datasets": {"test": {"name": "test", "type": "S2NAIPDataset", "phase": "test", "scale": 4, "n_s2_images": 8,
                                      "sentinel2_path": os.path.join(mini, "sentinel2"), "naip_path": os.path.join(mini, "naip")}},
           "network_g": dict(type="SSR_RRDBNet", **g_kw),
           "path": {"pretrain_network_g": str(tmp_path / "net_g.pth"), "param_key_g": "params_ema", "strict_load_g": True,
                    "visualization": str(tmp_path / "vis")},
           "test": {"save_img": True, "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 4, "test_y_channel": False},
                                                  "cpsnr": {"type": "calculate_cpsnr", "crop_border": 4, "test_y_channel": False}}}}
    res = test_pipeline(opt, log=lambda *_: None)
    assert set(res) == {"test"} and set(res["test"]) == {"psnr", "cpsnr"}
    # the metrics are those of the written images (CPU restatement on the PNG pairs)
    from PIL import Image
    import numpy as np
    from oracle import metrics_oracle as MO
    files = sorted(os.listdir(tmp_path / "vis" / "test"))
    assert len(files) == 10
    ps, cs = [], []
    for idx in range(5):
        sr = np.asarray(Image.open(tmp_path / "vis" / "test" / f"{idx}_t.png"))
        gt = np.asarray(Image.open(tmp_path / "vis" / "test" / f"{idx}_t_gt.png"))
        assert sr.shape == gt.shape == (128, 128, 3)
        ps.append(MO.calculate_psnr(sr, gt, 4))
        cs.append(MO.calculate_cpsnr(sr, gt, 4))
    assert res["test"]["psnr"] == pytest.approx(sum(ps) / 5, rel=1e-9) and res["test"]["cpsnr"] == pytest.approx(sum(cs) / 5, rel=1e-9)
    bad = dict(opt, test={"save_img": False, "metrics": {"lpips": {"type": "calculate_lpips", "lpips_model": "vgg"}}})
    with pytest.raises(NotImplementedError, match="calculate_lpips"):
        test_pipeline(bad, log=lambda *_: None)


@pytest.mark.parametrize("name", ["stepref_plain", "stepref_feedlr_oldhr", "stepref_gated", "stepref_usm"])
def test_model_plugin_against_the_unmodified_reference_method(tmp_path, name):
    """The MODEL_REGISTRY plugin driven like ssr/train.py drives the reference model (feed_data(uint8 batch) ->
    optimize_parameters(it) -> get_current_log(); test()) against fixtures produced by EXECUTING the unmodified
    SSRESRGANModel.feed_data / optimize_parameters / test of /root/reference/ssr/models/ssr_esrgan_model.py:104-244
    (oracle/make_golden_refstep.py)."""
    from satlas_super_resolution_amd import models  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    fx = load_golden(name)
    opt = _opt(tmp_path, fx, feed_disc_lr=bool(fx["opt"].get("feed_disc_lr", False)), l1_gt_usm=bool(fx["opt"].get("l1_gt_usm", False)),
               gan_gt_usm=bool(fx["opt"].get("gan_gt_usm", False)))
    opt["train"].update({"ema_decay": fx["ema_decay"], "net_d_iters": fx["net_d_iters"], "net_d_init_iters": fx["net_d_init_iters"],
                         "optim_d": {"type": "Adam", "lr": fx["lr"], "weight_decay": 0, "betas": list(fx["betas"])},
                         "optim_g": {"type": "Adam", "lr": fx["lr"], "weight_decay": 0, "betas": list(fx["betas"])}})
    opt["train"].pop("scheduler", None)
    torch.save({"params": fx["g0"], "params_ema": fx["g0"]}, tmp_path / "g0.pth")
    torch.save({"params": fx["d0"]}, tmp_path / "d0.pth")
    opt["path"].update({"pretrain_network_g": str(tmp_path / "g0.pth"), "pretrain_network_d": str(tmp_path / "d0.pth"),
                        "param_key_g": "params", "strict_load_g": True})
    m = build_model(opt)
    for it, batch in enumerate(fx["data"], start=1):
        m.feed_data(batch)
        m.optimize_parameters(it)
        log, ref = m.get_current_log(), fx["logs"][it - 1]
        for k, v in ref.items():
            assert abs(log[k] - v) <= 1e-3 * max(1.0, abs(v)), (it, k, log[k], v)
        if "l_g_pix" not in ref:                  # a gated iteration: the reference's log has no generator terms
            assert log.get("l_g_pix", 0.0) == 0.0
    n = len(fx["data"])
    steps = fx["lr"] * n if name == "stepref_usm" else None      # the smooth targets of this fixture leave a few gradients at rounding level
    for k, v in fx["g_final"].items():
        _close_update(m.ts.g_store.tensor(k).cpu(), v, fx["g0"][k], ("G", k), 5e-2, steps)
    sd_d = m.ts.d_store.state_dict()
    for k, v in fx["d_final"].items():
        if k.endswith("_u") or k.endswith("_v"):
            assert rel_err(sd_d[k].cpu(), v) < 1e-3, ("D buffer", k)
        else:
            _close_update(sd_d[k].cpu(), v, fx["d0"][k], ("D", k), 5e-2, steps)
    ema = m.ts.ema_state_dict()
    for k, v in fx["g_ema_final"].items():
        _close_update(ema[k].cpu(), v, fx["g0"][k], ("EMA", k), 5e-2, None if steps is None else steps * (1 - fx["ema_decay"]) * n)
    if name == "stepref_usm":                     # the device's sharpened L1 target (ssr_usm_sharp) against what the reference run fed its L1 loss
        tgt = m.ts.l1_tgt[..., :3].permute(0, 3, 1, 2).float().cpu()
        assert m.ts.l1_tgt is not m.ts.real_in
        assert rel_err(tgt, fx["gt_usm_last"]) < 1e-5, rel_err(tgt, fx["gt_usm_last"])
    m.test()                                      # :235-244: net_g_ema under no_grad on the last batch
    assert parity_close(m.output.cpu(), fx["test_output"]), rel_err(m.output.cpu(), fx["test_output"])


@pytest.mark.parametrize("name", ["stepref_plain", "stepref_feedlr_oldhr"])
def test_model_plugin_in_its_default_arithmetic_against_the_unmodified_reference_method(tmp_path, name):
    """The same fixtures (executions of the unmodified SSRESRGANModel methods, oracle/make_golden_refstep.py) with the plugin left at its DEFAULT
    arithmetic - fp32h since round 6 (fp16-split forward, split-bf16 backward): the logged losses at 1e-3, the test() output at the gate, and the
    post-step parameters on the update with the noise-robust criterion (Adam turns a gradient at rounding level into a +-lr step in ANY fp32
    implementation: at most 0.1 % of a tensor's elements outside the tight bound, none by more than the maximal Adam step)."""
    from satlas_super_resolution_amd import models  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    fx = load_golden(name)
    opt = _opt(tmp_path, fx, feed_disc_lr=bool(fx["opt"].get("feed_disc_lr", False)), l1_gt_usm=bool(fx["opt"].get("l1_gt_usm", False)),
               gan_gt_usm=bool(fx["opt"].get("gan_gt_usm", False)))
    del opt["compute_dtype"]                                      # the plugin's own default
    opt["train"].update({"ema_decay": fx["ema_decay"], "net_d_iters": fx["net_d_iters"], "net_d_init_iters": fx["net_d_init_iters"],
                         "optim_d": {"type": "Adam", "lr": fx["lr"], "weight_decay": 0, "betas": list(fx["betas"])},
                         "optim_g": {"type": "Adam", "lr": fx["lr"], "weight_decay": 0, "betas": list(fx["betas"])}})
    opt["train"].pop("scheduler", None)
    torch.save({"params": fx["g0"], "params_ema": fx["g0"]}, tmp_path / "g0.pth")
    torch.save({"params": fx["d0"]}, tmp_path / "d0.pth")
    opt["path"].update({"pretrain_network_g": str(tmp_path / "g0.pth"), "pretrain_network_d": str(tmp_path / "d0.pth"),
                        "param_key_g": "params", "strict_load_g": True})
    m = build_model(opt)
    assert m.compute_dtype == "fp32h"
    for it, batch in enumerate(fx["data"], start=1):
        m.feed_data(batch)
        m.optimize_parameters(it)
        log, ref = m.get_current_log(), fx["logs"][it - 1]
        for k, v in ref.items():
            assert abs(log[k] - v) <= 1e-3 * max(1.0, abs(v)), (it, k, log[k], v)
    steps = fx["lr"] * len(fx["data"])
    for k, v in fx["g_final"].items():
        _close_update(m.ts.g_store.tensor(k).cpu(), v, fx["g0"][k], ("G", k), 5e-2, steps)
    sd_d = m.ts.d_store.state_dict()
    for k, v in fx["d_final"].items():
        if k.endswith("_u") or k.endswith("_v"):
            assert rel_err(sd_d[k].cpu(), v) < 1e-3, ("D buffer", k)
        else:
            _close_update(sd_d[k].cpu(), v, fx["d0"][k], ("D", k), 5e-2, steps)
    m.test()
    assert parity_close(m.output.cpu(), fx["test_output"]), rel_err(m.output.cpu(), fx["test_output"])


def test_default_arithmetic_reports_an_activation_beyond_fp16_range_instead_of_logging_nan(tmp_path):
    """compute_dtype fp32h (the plugin's default): an input beyond fp16's 65504 makes the split forward non-finite - get_current_log() raises and
    names the mode to switch to; the same batch in fp32f trains"""
    from satlas_super_resolution_amd import models  # noqa: F401
    from satlas_super_resolution_amd.registry import build_model
    fx = load_golden("stepref_plain")
    for mode, ok in (("fp32h", False), ("fp32f", True)):
        opt = _opt(tmp_path, fx)
        opt["compute_dtype"] = mode
        opt["train"].pop("scheduler", None)
        m = build_model(opt)
        batch = fx["data"][0]
        m.feed_data(batch)
        m.ts.g_plan.xin[0, 0, 0, 0] = 1.0e5                      # one LR sample far outside any image: its hi piece overflows fp16 in conv_first
        m.optimize_parameters(1)
        if ok:
            assert all(v == v for v in m.get_current_log().values())
        else:
            with pytest.raises(FloatingPointError, match="fp32f"):
                m.get_current_log()
