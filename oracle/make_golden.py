"""TEST INFRASTRUCTURE — generates tests/golden/*.pt from the UNMODIFIED reference classes.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

Every fixture stores: the reference module's state_dict, the seeded inputs, and the outputs /
gradients / post-step state the *reference code* produced on CPU fp32.  tests/test_oracle_golden.py
checks oracle/esrgan_oracle.py against them (CPU), tests/test_gpu_parity.py checks the HIP path
against them (GPU).  The step fixture drives the reference nn.Modules with torch.optim.Adam and
nn.BCEWithLogitsLoss/L1Loss in the order of ssr/models/ssr_esrgan_model.py:119-233 (the model class
itself cannot be imported: it needs BasicSR's SRGANModel).
"""
import copy
import os
import sys
from collections import OrderedDict

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_shim import load_reference_archs, REFERENCE_ROOT  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def sd_of(m):
    return OrderedDict((k, v.detach().clone()) for k, v in m.state_dict().items())


def gen_generator(name, seed, B, H, W, **kw):
    G, _, _ = load_reference_archs()
    torch.manual_seed(seed)
    net = G(**kw).train()
    # give RDB biases non-zero values so that bias handling is exercised (reference init zeroes them)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith(".bias"):
                p.add_(torch.randn_like(p) * 0.05)
    sd = sd_of(net)
    x = torch.rand(B, kw["num_in_ch"], H, W)
    x.requires_grad_(True)
    y = net(x)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    fx = {"kwargs": kw, "state_dict": sd, "x": x.detach(), "y": y.detach(), "r": r,
          "dx": x.grad.detach(), "grads": OrderedDict((n, p.grad.detach().clone()) for n, p in net.named_parameters())}
    torch.save(fx, os.path.join(OUT, name + ".pt"))
    print(name, "y", tuple(y.shape), "abs max", float(y.abs().max()))


def gen_discriminator(name, seed, B, H, W, **kw):
    _, D, _ = load_reference_archs()
    torch.manual_seed(seed)
    net = D(**kw).train()
    sd0 = sd_of(net)
    x = torch.rand(B, kw["num_in_ch"], H, W, requires_grad=True)
    y = net(x)  # one power iteration happens inside (train mode)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    sd1 = sd_of(net)
    grads = OrderedDict((n, p.grad.detach().clone()) for n, p in net.named_parameters())
    # eval-mode forward with the updated buffers (no power iteration)
    net.eval()
    with torch.no_grad():
        y_eval = net(x.detach())
    fx = {"kwargs": kw, "state_dict_before": sd0, "state_dict_after": sd1, "x": x.detach(), "y": y.detach(),
          "r": r, "dx": x.grad.detach(), "grads": grads, "y_eval": y_eval}
    torch.save(fx, os.path.join(OUT, name + ".pt"))
    print(name, "y", tuple(y.shape), "abs max", float(y.abs().max()))


def gen_step(name, seed, B, n_iters, g_kw, d_kw, feed_disc_lr, l1_w=1.0, gan_w=0.1, lr=1e-4, betas=(0.9, 0.99),
             ema_decay=0.999):
    """Reference modules driven in the order of optimize_parameters (ssr_esrgan_model.py:119-233)."""
    G, D, _ = load_reference_archs()
    torch.manual_seed(seed)
    net_g = G(**g_kw).train()
    net_d = D(**d_kw).train()
    net_g_ema = copy.deepcopy(net_g).eval()
    opt_g = torch.optim.Adam(net_g.parameters(), lr=lr, weight_decay=0, betas=betas)
    opt_d = torch.optim.Adam(net_d.parameters(), lr=lr, weight_decay=0, betas=betas)
    bce = torch.nn.BCEWithLogitsLoss()
    g0, d0 = sd_of(net_g), sd_of(net_d)
    data, logs, g_grads1, d_grads1 = [], [], None, None
    for it in range(1, n_iters + 1):
        lr_img = torch.rand(B, g_kw["num_in_ch"], 8, 8)
        gt = torch.rand(B, 3, 32, 32)
        data.append((lr_img, gt))
        log = OrderedDict()
        lr_resized = F.interpolate(lr_img, scale_factor=4)
        for p in net_d.parameters():
            p.requires_grad = False
        opt_g.zero_grad()
        output = net_g(lr_img)
        l_g_pix = l1_w * F.l1_loss(output, gt)
        disc_in = torch.cat((output, lr_resized), 1) if feed_disc_lr else output
        fake_g_pred = net_d(disc_in)
        l_g_gan = gan_w * bce(fake_g_pred, torch.ones_like(fake_g_pred))
        (l_g_pix + l_g_gan).backward()
        if it == 1:
            g_grads1 = OrderedDict((n, p.grad.detach().clone()) for n, p in net_g.named_parameters())
        opt_g.step()
        for p in net_d.parameters():
            p.requires_grad = True
        real_in = torch.cat((gt, lr_resized), 1) if feed_disc_lr else gt
        fake_in = torch.cat((output, lr_resized), 1) if feed_disc_lr else output
        opt_d.zero_grad()
        real_d_pred = net_d(real_in)
        l_d_real = bce(real_d_pred, torch.ones_like(real_d_pred))
        l_d_real.backward()
        fake_d_pred = net_d(fake_in.detach().clone())
        l_d_fake = bce(fake_d_pred, torch.zeros_like(fake_d_pred))
        l_d_fake.backward()
        if it == 1:
            d_grads1 = OrderedDict((n, p.grad.detach().clone()) for n, p in net_d.named_parameters())
        opt_d.step()
        with torch.no_grad():
            ema_p = dict(net_g_ema.named_parameters())
            for k, p in net_g.named_parameters():
                ema_p[k].mul_(ema_decay).add_(p.data, alpha=1 - ema_decay)
        log["l_g_pix"], log["l_g_gan"] = l_g_pix.item(), l_g_gan.item()
        log["l_d_real"], log["out_d_real"] = l_d_real.item(), real_d_pred.detach().mean().item()
        log["l_d_fake"], log["out_d_fake"] = l_d_fake.item(), fake_d_pred.detach().mean().item()
        logs.append(log)
    fx = {"g_kwargs": g_kw, "d_kwargs": d_kw, "feed_disc_lr": feed_disc_lr, "l1_weight": l1_w, "gan_weight": gan_w,
          "lr": lr, "betas": betas, "ema_decay": ema_decay, "g0": g0, "d0": d0, "data": data, "logs": logs,
          "g_grads_iter1": g_grads1, "d_grads_iter1": d_grads1, "g_final": sd_of(net_g), "d_final": sd_of(net_d),
          "g_ema_final": sd_of(net_g_ema), "output_last": output.detach()}
    torch.save(fx, os.path.join(OUT, name + ".pt"))
    print(name, logs[-1])


def gen_index_maps():
    _, _, au = load_reference_archs()
    fx = {}
    for (c, hh, hw, s) in [(2, 4, 6, 2), (3, 8, 8, 4), (1, 2, 2, 2)]:
        x = torch.arange(c * hh * hw, dtype=torch.float32).view(1, c, hh, hw)
        fx[f"unshuffle_{c}_{hh}_{hw}_{s}"] = au.pixel_unshuffle(x, s).to(torch.int64)
    x = torch.arange(2 * 3 * 5, dtype=torch.float32).view(1, 2, 3, 5)
    fx["nearest2_2_3_5"] = F.interpolate(x, scale_factor=2, mode="nearest").to(torch.int64)
    fx["nearest4_2_3_5"] = F.interpolate(x, scale_factor=4).to(torch.int64)  # lr_resized, ssr_esrgan_model.py:133
    torch.save(fx, os.path.join(OUT, "index_maps.pt"))
    print("index_maps", list(fx))


def gen_infer_utils():
    """Golden vectors of the UNMODIFIED /root/reference/ssr/utils/infer_utils.py:format_s2naip_data (frame selection with
    `random.sample`, black-pixel rejection, /255).  The module imports skimage.io (absent here) only for `stitch`'s PNG
    I/O, so an empty stand-in module is enough to import it."""
    import importlib.util
    import random
    import sys
    import types
    import numpy as np
    sk, skio = types.ModuleType("skimage"), types.ModuleType("skimage.io")
    sk.io = skio
    sys.modules.setdefault("skimage", sk)
    sys.modules.setdefault("skimage.io", skio)
    spec = importlib.util.spec_from_file_location("ref_infer_utils", os.path.join(REFERENCE_ROOT, "ssr/utils/infer_utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.RandomState(7)
    cases = []
    for T, n, nblack in [(8, 8, 0), (12, 8, 3), (10, 8, 5), (6, 1, 2)]:
        data = rng.randint(1, 256, size=(T * 32, 32, 3)).astype(np.uint8)
        for f in rng.choice(T, nblack, replace=False):
            data[f * 32 + 5, 7, rng.randint(3)] = 0            # a single zero channel value marks the frame "bad" (:17)
        random.seed(1234 + T)
        t, img = m.format_s2naip_data(data, n, "cpu")
        cases.append({"data": torch.from_numpy(data), "n": n, "seed": 1234 + T, "tensor": t.clone(),
                      "image": torch.from_numpy(img.copy())})
    torch.save({"format_s2naip_data": cases}, os.path.join(OUT, "infer_utils.pt"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    torch.use_deterministic_algorithms(True)
    # generator: ragged sizes, odd channel counts (exercise channel padding), scale variants
    gen_generator("g_tiny_ragged", 11, B=2, H=12, W=10, num_in_ch=5, num_out_ch=3, scale=4, num_feat=16,
                  num_block=2, num_grow_ch=8)
    gen_generator("g_mid_24ch", 12, B=1, H=16, W=16, num_in_ch=24, num_out_ch=3, scale=4, num_feat=32,
                  num_block=1, num_grow_ch=16)
    gen_generator("g_scale2", 13, B=1, H=8, W=12, num_in_ch=3, num_out_ch=3, scale=2, num_feat=16,
                  num_block=1, num_grow_ch=8)
    gen_generator("g_scale1", 14, B=1, H=8, W=8, num_in_ch=2, num_out_ch=3, scale=1, num_feat=16,
                  num_block=1, num_grow_ch=8)
    gen_discriminator("d_tiny", 21, B=2, H=16, W=24, num_in_ch=3, num_feat=8, skip_connection=True)
    gen_discriminator("d_in6_noskip", 22, B=1, H=32, W=32, num_in_ch=6, num_feat=8, skip_connection=False)
    gen_step("step_tiny", 31, B=2, n_iters=2,
             g_kw=dict(num_in_ch=6, num_out_ch=3, scale=4, num_feat=16, num_block=1, num_grow_ch=8),
             d_kw=dict(num_in_ch=3, num_feat=8, skip_connection=True), feed_disc_lr=False)
    gen_step("step_tiny_feedlr", 32, B=1, n_iters=2,
             g_kw=dict(num_in_ch=6, num_out_ch=3, scale=4, num_feat=16, num_block=1, num_grow_ch=8),
             d_kw=dict(num_in_ch=9, num_feat=8, skip_connection=True), feed_disc_lr=True)
    gen_index_maps()
    gen_infer_utils()
