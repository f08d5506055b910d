"""The seam that no box of this build has ever exercised: the reference's OWN driver code resolving our classes through BasicSR.

`ssr/train.py:62` calls `basicsr.models.build_model(opt)`, which looks `opt['model_type']` up in BasicSR's MODEL_REGISTRY; the model
then calls `basicsr.archs.build_network(opt['network_g'])` (ARCH_REGISTRY).  satlas_super_resolution_amd/registry.py registers the HIP
classes into THOSE registries when BasicSR is importable.  Neither BasicSR (requirements.txt:1, basicsr==1.4.2) nor the reference
tree travels with this repo, so the test is gated:

    SSR_REFERENCE_ROOT=/path/to/satlas-super-resolution  (+ an installed basicsr)  pytest tests/test_reference_seam.py -m gpu

and is skipped everywhere else (build container: no BasicSR; GPU boxes: neither).  The two sides of the seam are pinned separately
all the same: our registry / model plugin against the unmodified `optimize_parameters` (tests/test_gpu_boundary.py), the key layout
against the reference modules' state_dicts (tests/test_host_boundary.py)."""
import importlib.util
import os
import sys

import pytest
import torch

REF = os.environ.get("SSR_REFERENCE_ROOT")
HAVE = bool(REF) and os.path.isdir(REF or "") and importlib.util.find_spec("basicsr") is not None
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not HAVE, reason="needs SSR_REFERENCE_ROOT and an installed BasicSR (see the module docstring)")]


def _opt(tmp_path):
    return {
        "name": "seam", "model_type": "SSRESRGANModel", "scale": 4, "num_gpu": 1, "manual_seed": 0, "is_train": True, "dist": False,
        "rank": 0, "world_size": 1, "l1_gt_usm": False, "percep_gt_usm": False, "gan_gt_usm": False, "feed_disc_lr": False,
        "network_g": {"type": "SSR_RRDBNet", "num_in_ch": 24, "num_out_ch": 3, "num_feat": 64, "num_block": 1, "num_grow_ch": 32},
        "network_d": {"type": "SSR_UNetDiscriminatorSN", "num_in_ch": 3, "num_feat": 64, "skip_connection": True},
        "path": {"pretrain_network_g": None, "models": str(tmp_path / "models"), "training_states": str(tmp_path / "states"),
                 "visualization": str(tmp_path / "vis"), "experiments_root": str(tmp_path), "log": str(tmp_path)},
        "train": {"ema_decay": 0.999, "optim_g": {"type": "Adam", "lr": 1e-4, "weight_decay": 0, "betas": [0.9, 0.99]},
                  "optim_d": {"type": "Adam", "lr": 1e-4, "weight_decay": 0, "betas": [0.9, 0.99]},
                  "scheduler": {"type": "MultiStepLR", "milestones": [400000], "gamma": 0.5}, "total_iter": 2, "warmup_iter": -1,
                  "pixel_opt": {"type": "L1Loss", "loss_weight": 1.0, "reduction": "mean"},
                  "gan_opt": {"type": "GANLoss", "gan_type": "vanilla", "real_label_val": 1.0, "fake_label_val": 0.0, "loss_weight": 0.1},
                  "net_d_iters": 1, "net_d_init_iters": 0},
        "logger": {"print_freq": 1, "save_checkpoint_freq": 100},
    }


def test_basicsr_build_model_resolves_the_hip_plugin_and_trains(tmp_path):
    import satlas_super_resolution_amd.archs  # noqa: F401  (registers SSR_RRDBNet / SSR_UNetDiscriminatorSN in BasicSR's ARCH_REGISTRY)
    import satlas_super_resolution_amd.models  # noqa: F401  (SSRESRGANModel in MODEL_REGISTRY)
    from basicsr.archs import build_network        # what ssr_esrgan_model.py:43,53 call
    from basicsr.models import build_model         # what ssr/train.py:62 calls
    from basicsr.utils.registry import ARCH_REGISTRY, MODEL_REGISTRY
    from satlas_super_resolution_amd.archs.rrdbnet_arch import SSR_RRDBNet
    from satlas_super_resolution_amd.models.ssr_esrgan_model import SSRESRGANModel
    assert ARCH_REGISTRY.get("SSR_RRDBNet") is SSR_RRDBNet and MODEL_REGISTRY.get("SSRESRGANModel") is SSRESRGANModel
    net = build_network({"type": "SSR_RRDBNet", "num_in_ch": 24, "num_out_ch": 3, "num_feat": 64, "num_block": 1, "num_grow_ch": 32})
    assert isinstance(net, SSR_RRDBNet)
    model = build_model(_opt(tmp_path))
    assert isinstance(model, SSRESRGANModel)
    torch.manual_seed(0)
    for it in (1, 2):                              # the loop body of ssr/train.py:106-110
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_data({"lr": torch.randint(0, 256, (2, 24, 32, 32), dtype=torch.uint8), "hr": torch.randint(0, 256, (2, 3, 128, 128), dtype=torch.uint8)})
        model.optimize_parameters(it)
    log = model.get_current_log()
    assert {"l_g_pix", "l_g_gan", "l_d_real", "l_d_fake"} <= set(log) and all(v == v for v in log.values())
    # and the reference's inference helper with the reference's own import line swapped
    sys.path.insert(0, REF)
    from satlas_super_resolution_amd.utils.model_utils import build_network as build_infer
    g = build_infer({"n_lr_images": 8, "network_g": {"type": "SSR_RRDBNet", "num_in_ch": 24, "num_out_ch": 3, "num_feat": 64, "num_block": 1, "num_grow_ch": 32}})
    assert isinstance(g, SSR_RRDBNet)
