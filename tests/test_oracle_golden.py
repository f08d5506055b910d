"""CPU: pins oracle/esrgan_oracle.py (the restatement) against the golden vectors the unmodified
reference classes produced (oracle/make_golden.py).  No GPU, no /root/reference needed."""
import copy
from collections import OrderedDict

import pytest
import math

import torch

from conftest import load_golden, rel_err
from oracle import esrgan_oracle as O

TOL = 2e-5  # fp32 CPU vs fp32 CPU: only summation order differs


@pytest.mark.parametrize("name", ["g_tiny_ragged", "g_mid_24ch", "g_scale2", "g_scale1"])
def test_generator_forward_backward(name):
    fx = load_golden(name)
    sd = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in fx["state_dict"].items())
    x = fx["x"].clone().requires_grad_(True)
    y = O.generator_forward(sd, x, fx["kwargs"]["scale"])
    assert y.shape == fx["y"].shape
    assert rel_err(y, fx["y"]) < TOL
    (y * fx["r"]).sum().backward()
    assert rel_err(x.grad, fx["dx"]) < TOL
    for k, g in fx["grads"].items():
        assert rel_err(sd[k].grad, g) < 5e-5, k


@pytest.mark.parametrize("name", ["d_tiny", "d_in6_noskip"])
def test_discriminator_forward_backward(name):
    fx = load_golden(name)
    sd = OrderedDict((k, (v.clone().requires_grad_(True) if k in O.D_PARAM_KEYS else v.clone()))
                     for k, v in fx["state_dict_before"].items())
    x = fx["x"].clone().requires_grad_(True)
    y = O.discriminator_forward(sd, x, train=True, skip_connection=fx["kwargs"]["skip_connection"])
    assert rel_err(y, fx["y"]) < TOL
    (y * fx["r"]).sum().backward()
    assert rel_err(x.grad, fx["dx"]) < TOL
    for k, g in fx["grads"].items():
        assert rel_err(sd[k].grad, g) < 5e-5, k
    # power-iteration buffers after one train-mode forward
    for n in O.SN_LAYERS:
        for s in (".weight_u", ".weight_v"):
            assert rel_err(sd[n + s], fx["state_dict_after"][n + s]) < TOL, n + s
    # eval: no power iteration, uses the stored u, v
    sd_eval = OrderedDict((k, v.clone()) for k, v in fx["state_dict_after"].items())
    with torch.no_grad():
        y_eval = O.discriminator_forward(sd_eval, fx["x"], train=False,
                                         skip_connection=fx["kwargs"]["skip_connection"])
    assert rel_err(y_eval, fx["y_eval"]) < TOL
    for n in O.SN_LAYERS:
        assert torch.equal(sd_eval[n + ".weight_u"], fx["state_dict_after"][n + ".weight_u"])


@pytest.mark.parametrize("name", ["step_tiny", "step_tiny_feedlr"])
def test_train_step(name):
    fx = load_golden(name)
    cfg = O.StepConfig(l1_weight=fx["l1_weight"], gan_weight=fx["gan_weight"], lr_g=fx["lr"], lr_d=fx["lr"],
                       betas=tuple(fx["betas"]), ema_decay=fx["ema_decay"], feed_disc_lr=fx["feed_disc_lr"])
    m = O.ESRGANOracle(fx["g0"], fx["d0"], cfg)
    for it, (lr, gt) in enumerate(fx["data"], start=1):
        log = m.step(lr, gt, it)
        for k, v in fx["logs"][it - 1].items():
            assert abs(log[k] - v) <= 2e-5 * max(1.0, abs(v)), (it, k, log[k], v)
        if it == 1:
            for k, g in fx["g_grads_iter1"].items():
                assert rel_err(m.g_grads[k], g) < 1e-4, k
            for k, g in fx["d_grads_iter1"].items():
                assert rel_err(m.d_grads[k], g) < 1e-4, k
    # Adam with tiny grads amplifies rounding through g/sqrt(v): compare the parameter *update*
    for k, v in fx["g_final"].items():
        upd_ref = v - fx["g0"][k]
        upd = m.g[k] - fx["g0"][k]
        assert (upd - upd_ref).abs().max() <= 2e-2 * upd_ref.abs().max() + 1e-9, k
    for k, v in fx["d_final"].items():
        ref0 = fx["d0"][k]
        assert (m.d[k] - v).abs().max() <= 2e-2 * (v - ref0).abs().max() + 1e-6, k
    for k, v in fx["g_ema_final"].items():
        assert rel_err(m.g_ema[k], v) < 1e-5, k
    assert rel_err(m.output, fx["output_last"]) < 1e-4


@pytest.mark.parametrize("name", ["stepref_plain", "stepref_feedlr_oldhr", "stepref_gated", "stepref_usm"])
def test_train_step_against_the_unmodified_reference_method(name):
    """The oracle's step against fixtures produced by EXECUTING the reference's own SSRESRGANModel.feed_data /
    optimize_parameters / test (ssr_esrgan_model.py:104-244; oracle/make_golden_refstep.py) — not a re-typed loop: freeze /
    unfreeze, detach().clone(), the two D backwards, the gate, the discriminator-input channel order and the EMA placement are
    the reference's text."""
    fx = load_golden(name)
    cfg = O.StepConfig(l1_weight=fx["l1_weight"], gan_weight=fx["gan_weight"], lr_g=fx["lr"], lr_d=fx["lr"],
                       betas=tuple(fx["betas"]), ema_decay=fx["ema_decay"], feed_disc_lr=bool(fx["opt"].get("feed_disc_lr", False)),
                       net_d_iters=fx["net_d_iters"], net_d_init_iters=fx["net_d_init_iters"],
                       l1_gt_usm=bool(fx["opt"].get("l1_gt_usm", False)), gan_gt_usm=bool(fx["opt"].get("gan_gt_usm", False)))
    if name == "stepref_usm":
        # stepref_usm: the shipped setting l1_gt_usm = True (esrgan_s2naip_urban.yml:9) through the unmodified feed_data /
        # optimize_parameters with BasicSR's USMSharp text: the sharpened target itself, then the step that uses it
        gt_last = fx["data"][-1]["hr"].float() / 255
        assert float((fx["gt_usm_last"] - gt_last).abs().mean()) > 1e-3            # the sharpener does something on this image
        assert rel_err(O.usm_sharp(gt_last), fx["gt_usm_last"]) < 1e-6
    m = O.ESRGANOracle(fx["g0"], fx["d0"], cfg)
    for it, batch in enumerate(fx["data"], start=1):
        lr, gt = batch["lr"].float() / 255, batch["hr"].float() / 255            # feed_data :107-109
        old = batch["old_hr"].float() / 255 if "old_hr" in batch else None
        log = m.step(lr, gt, it, old_hr=old)
        ref = fx["logs"][it - 1]
        g_on = it % fx["net_d_iters"] == 0 and it > fx["net_d_init_iters"]
        assert ("l_g_pix" in ref) == g_on == ("l_g_pix" in log), (it, list(ref), list(log))
        for k, v in ref.items():
            assert abs(log[k] - v) <= 2e-5 * max(1.0, abs(v)), (it, k, log[k], v)
        if fx["g_grads_first"] is not None and it == fx["g_grads_first"][0]:
            for k, g in fx["g_grads_first"][1].items():
                assert rel_err(m.g_grads[k], g) < 1e-4, k
        if it == 1:
            for k, g in fx["d_grads_iter1"].items():
                assert rel_err(m.d_grads[k], g) < 1e-4, k
    for k, v in fx["g_final"].items():
        upd_ref, upd = v - fx["g0"][k], m.g[k] - fx["g0"][k]
        assert (upd - upd_ref).abs().max() <= 2e-2 * upd_ref.abs().max() + 1e-9, k
    for k, v in fx["d_final"].items():
        assert (m.d[k] - v).abs().max() <= 2e-2 * (v - fx["d0"][k]).abs().max() + 1e-6, k
    for k, v in fx["g_ema_final"].items():
        assert rel_err(m.g_ema[k], v) < 1e-5, k
    # test(): net_g_ema on the last lr batch
    with torch.no_grad():
        out = O.generator_forward(m.g_ema, fx["data"][-1]["lr"].float() / 255, 4)
    assert rel_err(out, fx["test_output"]) < 1e-4


def test_index_maps():
    fx = load_golden("index_maps")
    for key, ref in fx.items():
        if key.startswith("unshuffle"):
            c, hh, hw, s = map(int, key.split("_")[1:])
            x = torch.arange(c * hh * hw, dtype=torch.float32).view(1, c, hh, hw)
            assert torch.equal(O.pixel_unshuffle(x, s).to(torch.int64), ref)
    x = torch.arange(2 * 3 * 5, dtype=torch.float32).view(1, 2, 3, 5)
    for f, key in ((2, "nearest2_2_3_5"), (4, "nearest4_2_3_5")):
        iy = torch.arange(3 * f) // f
        ix = torch.arange(5 * f) // f
        assert torch.equal(x[:, :, iy][:, :, :, ix].to(torch.int64), fx[key])
    assert torch.equal(O.nearest_up2_index(6), torch.tensor([0, 0, 1, 1, 2, 2]))
    offs = O.stitch_offsets(16, 128)
    assert offs[3][5] == (384, 640) and offs[15][15] == (1920, 1920)
    q = O.quantize_u8_truncate(torch.tensor([-0.1, 0.0, 0.5, 0.999, 1.0, 1.5]))
    assert q.tolist() == [0, 0, 127, 254, 255, 255]


def test_flop_model_matches_baseline_md():
    # BASELINE.md §3 table
    assert abs(O.step_gflop_per_image(3, 3) - 213.71) < 0.02
    assert abs(O.step_gflop_per_image(24, 3) - 213.76) < 0.02
    assert abs(O.step_gflop_per_image(96, 3) - 213.93) < 0.02
    assert abs(O.step_gflop_per_image(24, 27) - 216.48) < 0.02
    assert abs(2 * sum(O.generator_conv_macs(3).values()) / 1e9 - 36.714) < 0.005
    assert abs(2 * sum(O.discriminator_conv_macs(3).values()) / 1e9 - 12.960) < 0.005


def test_usm_sharp_oracle_properties():
    """BasicSR's USMSharp restated in oracle/esrgan_oracle.py (parity unpinned by the reference: the class lives in the
    absent basicsr package; anchored on ssr_esrgan_model.py:31,109).  Known answers: OpenCV's sigma rule for ksize 51,
    normalisation, identity on flat images, clipping and the soft blend on a step edge."""
    from oracle import esrgan_oracle as O
    k = O.usm_gaussian_kernel1d(50, 0)
    assert k.numel() == 51 and abs(float(k.sum()) - 1.0) < 1e-12
    assert torch.allclose(k, k.flip(0))
    assert abs(float(k[25] / k[24]) - math.exp(1.0 / 128.0)) < 1e-12          # sigma = 8
    flat = torch.full((1, 3, 64, 64), 0.37)
    assert torch.allclose(O.usm_sharp(flat), flat, atol=1e-6)
    edge = torch.zeros(1, 1, 64, 64)
    edge[..., 32:] = 0.8
    out = O.usm_sharp(edge)
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert float(out[0, 0, 32, 33]) > 0.8 + 0.05            # overshoot on the bright side of the edge
    assert float(out[0, 0, 32, 30]) == 0.0                  # dark side: sharp value clipped at 0
    assert torch.allclose(out[0, 0, :, 0], edge[0, 0, :, 0])   # far from the edge: residual below threshold


def test_metrics_oracle_cpsnr_matches_reference_golden():
    """oracle/metrics_oracle.calculate_cpsnr against values of the unmodified /root/reference/ssr/metrics/cpsnr.py
    (oracle/make_metric_golden.py); PSNR / SSIM known answers (BasicSR pieces: parity unpinned by the reference)."""
    import numpy as np
    from oracle import metrics_oracle as MO
    for c in load_golden("cpsnr"):
        a, b = c["img"].numpy(), c["img2"].numpy()
        v = MO.calculate_cpsnr(a, b, c["crop_border"])
        assert abs(v - c["value"]) <= 1e-9 * max(1.0, abs(c["value"])), (v, c["value"])
    rng = np.random.RandomState(0)
    a = rng.randint(0, 256, (32, 32, 3)).astype(np.uint8)
    assert MO.calculate_psnr(a, a, 4) == float("inf") and abs(MO.calculate_ssim(a, a, 4) - 1.0) < 1e-12
    b = a.copy()
    b[4:-4, 4:-4] = np.clip(a[4:-4, 4:-4].astype(int) + 5, 0, 255).astype(np.uint8)   # +5 wherever it does not saturate
    d = (a[4:-4, 4:-4].astype(float) - b[4:-4, 4:-4]) ** 2
    assert abs(MO.calculate_psnr(a, b, 4) - 10 * np.log10(255.0 ** 2 / d.mean())) < 1e-12
    assert 0.9 < MO.calculate_ssim(a, b, 4) < 1.0
    g = MO._gauss()
    assert abs(g.sum() - 1) < 1e-15 and abs(g[5] / g[4] - np.exp(1 / (2 * 1.5 ** 2))) < 1e-12


def _full_inputs(fx, shape):
    g = torch.Generator().manual_seed(fx["seed"] + 2)
    x = torch.rand(*shape, generator=g)
    assert torch.equal(x[0, :, 0, 0], fx["x_check"]), "seeded input differs from the one the golden was generated with"
    return x, g


@pytest.mark.parametrize("name,c_in", [("full_g24", 24), ("full_g96", 96)])
def test_generator_full_size_matches_reference_golden(name, c_in):
    """The oracle at the BENCHMARKED architecture (nf=64, gc=32, nb=23) against the unmodified reference class
    (oracle/make_golden_fullsize.py; parameters rebuilt from the seed): output, input gradient, parameter gradients."""
    from oracle.make_golden_fullsize import biased
    fx = load_golden(name)
    sd = biased(O.generator_init(seed=fx["seed"], **fx["kwargs"]), fx["seed"] + 1)
    x, g = _full_inputs(fx, (2, c_in, 32, 32))
    sdg = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    x = x.requires_grad_(True)
    y = O.generator_forward(sdg, x, 4)
    assert rel_err(y, fx["y"]) < TOL
    r = torch.randn(y.shape, generator=g)
    (y * r).sum().backward()
    assert rel_err(x.grad, fx["dx"]) < 1e-4
    for k, gr in fx["grads"].items():
        assert rel_err(sdg[k].grad, gr) < 1e-4, k


@pytest.mark.parametrize("name", ["full_d3", "full_d27"])
def test_discriminator_full_size_matches_reference_golden(name):
    fx = load_golden(name)
    sd = O.discriminator_init(fx["c_d"], 64, seed=fx["seed"])
    sdg = OrderedDict((k, (v.clone().requires_grad_(True) if k in O.D_PARAM_KEYS else v.clone())) for k, v in sd.items())
    x, g = _full_inputs(fx, (1, fx["c_d"], 128, 128))
    x = x.requires_grad_(True)
    y = O.discriminator_forward(sdg, x, train=True)
    assert rel_err(y, fx["y"]) < TOL
    r = torch.randn(y.shape, generator=g)
    (y * r).sum().backward()
    assert rel_err(x.grad[:, :3], fx["dx_first3"]) < 1e-4 and rel_err(x.grad[:, -3:], fx["dx_last3"]) < 1e-4
    for k, gr in fx["grads"].items():
        assert rel_err(sdg[k].grad, gr) < 1e-4, k
    for k, v in fx["uv_after"].items():
        assert rel_err(sdg[k], v) < TOL, k
