"""TEST INFRASTRUCTURE — golden values of calculate_cpsnr from the UNMODIFIED reference function
(/root/reference/ssr/metrics/cpsnr.py:8-59), executed in the build container.  Writes tests/golden/cpsnr.pt
(seeded uint8 image pairs + the reference's values).  Run:  python oracle/make_metric_golden.py"""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402


def load_reference_cpsnr():
    ref_shim._install_stubs()
    ref_shim.load_reference_archs()          # installs the `ssr` package object pointing into the reference tree
    for sub in ("metrics", "utils"):         # skip ssr/metrics/__init__.py (pulls lpips / clip, absent here)
        name = f"ssr.{sub}"
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(ref_shim.REFERENCE_ROOT, "ssr", sub)]
            sys.modules[name] = m
    return importlib.import_module("ssr.metrics.cpsnr").calculate_cpsnr


def main():
    f = load_reference_cpsnr()
    rng = np.random.RandomState(7)
    cases = []
    for (h, w, crop, kind) in [(40, 40, 4, "noise"), (64, 48, 4, "shift"), (32, 32, 0, "bias"), (128, 128, 4, "blur")]:
        a = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        if kind == "noise":
            b = np.clip(a.astype(np.int32) + rng.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
        elif kind == "shift":      # b is a translated copy: one of the 81 offsets aligns them
            b = np.roll(a, (3, -2), axis=(0, 1))
        elif kind == "bias":
            b = np.clip(a.astype(np.int32) // 2 + 40, 0, 255).astype(np.uint8)
        else:
            b = ((a.astype(np.float32) + np.roll(a, 1, 0) + np.roll(a, 1, 1)) / 3).astype(np.uint8)
        cases.append({"img": torch.from_numpy(a), "img2": torch.from_numpy(b), "crop_border": crop,
                      "value": float(f(a, b, crop))})
    dst = os.path.join(ROOT, "tests", "golden", "cpsnr.pt")
    torch.save(cases, dst)
    print("wrote", dst, [c["value"] for c in cases])


if __name__ == "__main__":
    main()
