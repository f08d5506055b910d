"""Arch plugins, discovered by the `*_arch.py` filename suffix like /root/reference/ssr/archs/__init__.py:7-10."""
import importlib
import os

_here = os.path.dirname(os.path.abspath(__file__))
_arch_modules = [importlib.import_module(f"{__name__}.{f[:-3]}") for f in sorted(os.listdir(_here))
                 if f.endswith("_arch.py")]
