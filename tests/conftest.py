import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name + ".pt"), map_location="cpu", weights_only=False)


def parity_close(a, ref, rtol=1e-3, atol_frac=1e-3):
    """SURVEY.md §0 D5 / north_star: |a-ref| <= 1e-3*max|ref| + 1e-3*|ref| (fp32)."""
    import torch
    a, ref = a.detach().float().cpu(), ref.detach().float().cpu()
    scale = float(ref.abs().max())
    return torch.allclose(a, ref, rtol=rtol, atol=atol_frac * scale + 1e-30)


def rel_err(a, ref):
    a, ref = a.detach().float().cpu(), ref.detach().float().cpu()
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.fixture
def golden():
    return load_golden
