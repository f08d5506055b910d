"""TEST INFRASTRUCTURE — not product code.

Imports the *unmodified* reference arch classes from /root/reference so that the
CPU restatement in `oracle/esrgan_oracle.py` can be pinned against the real thing
and golden vectors can be generated (`oracle/make_golden.py`).

/root/reference exists only in the build container, never on the GPU box, so
nothing under tests/ -m gpu, bench.py or smoke() may call this module.

The reference's arch files need a handful of names from packages that are not
installed here (basicsr 1.4.2, torchvision).  We provide inert
in-memory stand-ins for exactly those names:

  basicsr.utils.registry.ARCH_REGISTRY   (rrdbnet_arch.py:10, discriminator_arch.py:9)
  basicsr.utils.get_root_logger/scandir  (arch_util.py module imports)
  basicsr.ops.dcn.*                      (arch_util.py:19-20, dead code on this path)
  torchvision                            (arch_util.py:8, dead code on this path)
"""
import importlib
import logging
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SSR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ssr", "archs"))


class _Registry:
    def __init__(self, name):
        self.name = name
        self._map = {}

    def register(self, obj=None):
        def deco(o):
            self._map[o.__name__] = o
            return o
        return deco if obj is None else deco(obj)

    def get(self, name):
        return self._map[name]


def _install_stubs():
    import torch.nn as nn

    if "basicsr" in sys.modules and not getattr(sys.modules["basicsr"], "_ssr_oracle_stub", False):
        return  # a real basicsr is installed: use it
    basicsr = types.ModuleType("basicsr")
    basicsr._ssr_oracle_stub = True
    utils = types.ModuleType("basicsr.utils")
    utils.get_root_logger = lambda *a, **k: logging.getLogger("basicsr")

    def scandir(path, suffix=None, recursive=False, full_path=False):
        for f in sorted(os.listdir(path)):
            if suffix is None or f.endswith(suffix):
                yield os.path.join(path, f) if full_path else f
    utils.scandir = scandir
    registry = types.ModuleType("basicsr.utils.registry")
    for n in ("ARCH", "MODEL", "LOSS", "METRIC", "DATASET"):
        setattr(registry, f"{n}_REGISTRY", _Registry(n.lower()))
    utils.registry = registry
    ops = types.ModuleType("basicsr.ops")
    dcn = types.ModuleType("basicsr.ops.dcn")

    class ModulatedDeformConvPack(nn.Module):  # only subclassed by dead DCNv2Pack
        pass
    dcn.ModulatedDeformConvPack = ModulatedDeformConvPack
    dcn.modulated_deform_conv = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    ops.dcn = dcn
    basicsr.utils, basicsr.ops = utils, ops
    sys.modules.update({
        "basicsr": basicsr, "basicsr.utils": utils, "basicsr.utils.registry": registry,
        "basicsr.ops": ops, "basicsr.ops.dcn": dcn,
    })
    if "torchvision" not in sys.modules:
        try:
            importlib.import_module("torchvision")
        except Exception:
            tv = types.ModuleType("torchvision")
            tv.__version__ = "0.16.0"
            sys.modules["torchvision"] = tv


def load_reference_archs():
    """Returns (SSR_RRDBNet, SSR_UNetDiscriminatorSN, arch_util module) — the reference's own classes."""
    if not reference_available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    _install_stubs()
    # package objects pointing into the reference tree; skip ssr/archs/__init__.py
    # (it would import srcnn_arch -> kornia, absent).
    if "ssr" not in sys.modules or not hasattr(sys.modules["ssr"], "_ssr_oracle_pkg"):
        ssr = types.ModuleType("ssr")
        ssr.__path__ = [os.path.join(REFERENCE_ROOT, "ssr")]
        ssr._ssr_oracle_pkg = True
        archs = types.ModuleType("ssr.archs")
        archs.__path__ = [os.path.join(REFERENCE_ROOT, "ssr", "archs")]
        ssr.archs = archs
        sys.modules["ssr"] = ssr
        sys.modules["ssr.archs"] = archs
    g = importlib.import_module("ssr.archs.rrdbnet_arch")
    d = importlib.import_module("ssr.archs.discriminator_arch")
    au = importlib.import_module("ssr.archs.arch_util")
    return g.SSR_RRDBNet, d.SSR_UNetDiscriminatorSN, au
