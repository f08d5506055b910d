"""TEST / MEASUREMENT INFRASTRUCTURE — times the UNMODIFIED reference on the CPU of the BUILD container (the reference cannot travel
to the GPU box): SSRESRGANModel.optimize_parameters (ssr_esrgan_model.py:119-233) driving the reference's own SSR_RRDBNet(nf=64, nb=23,
gc=32) and SSR_UNetDiscriminatorSN(nf=64) at the metric's shapes (8 x S2 = 24 input channels, 32x32 -> 128x128), fp32, reduced batch;
plus the forward-only case of BASELINE.json configs[0].  Writes profiles/cpu_reference_step.json, which bench.py attaches to its
`cpu_baseline` record as `reference_in_build_container` ("the unmodified reference classes ... timed beside it").

    python -m oracle.time_reference_cpu [--batch 4] [--steps 3]
"""
import argparse
import json
import os
import time

import torch

from oracle.make_golden_refstep import _GANLoss, _L1Loss, load_reference_model_class


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    n_threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(n_threads)
    Model, G, D, Sharp = load_reference_model_class()
    torch.manual_seed(0)
    m = object.__new__(Model)
    m.device = torch.device("cpu")
    m.opt = {"l1_gt_usm": False, "percep_gt_usm": False, "gan_gt_usm": False}
    m.net_g = G(num_in_ch=24, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32).train()
    m.net_d = D(num_in_ch=3, num_feat=64, skip_connection=True).train()
    import copy
    m.net_g_ema = copy.deepcopy(m.net_g).eval()
    m.usm_sharpener = Sharp()
    m.cri_pix, m.cri_gan = _L1Loss(1.0), _GANLoss(0.1)
    m.cri_ldl = m.cri_perceptual = m.ssim_loss = m.clip_sim = None
    m.net_d_iters, m.net_d_init_iters, m.ema_decay = 1, 0, 0.999
    m.optimizer_g = torch.optim.Adam(m.net_g.parameters(), lr=1e-4, betas=(0.9, 0.99))
    m.optimizer_d = torch.optim.Adam(m.net_d.parameters(), lr=1e-4, betas=(0.9, 0.99))
    B = args.batch
    batch = {"lr": torch.randint(0, 256, (B, 24, 32, 32), dtype=torch.uint8), "hr": torch.randint(0, 256, (B, 3, 128, 128), dtype=torch.uint8)}
    m.feed_data(batch)
    m.optimize_parameters(1)          # warm-up
    ts = []
    for it in range(args.steps):
        t0 = time.perf_counter()
        m.optimize_parameters(it + 2)
        ts.append(time.perf_counter() - t0)
    t_step = sorted(ts)[len(ts) // 2]
    # configs[0]: 1 x S2 RGB forward only, batch 4
    g3 = G(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32).eval()
    x = torch.rand(4, 3, 32, 32)
    with torch.no_grad():
        g3(x)
        tf = []
        for _ in range(3):
            t0 = time.perf_counter()
            g3(x)
            tf.append(time.perf_counter() - t0)
    t_fwd = sorted(tf)[1]
    cpu = "unknown"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
    out = {"what": "unmodified reference classes + unmodified SSRESRGANModel.optimize_parameters on the BUILD container's CPU (not the GPU box's host)",
           "step_images_per_s": B / t_step, "step_seconds": t_step, "step_batch": B, "step_samples": ts,
           "forward_cfg0_images_per_s": 4 / t_fwd, "forward_cfg0_seconds": t_fwd,
           "threads": n_threads, "cpu": cpu, "dtype": "fp32", "torch": torch.__version__}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "cpu_reference_step.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
