"""Model plugins, discovered by the `*_model.py` suffix like /root/reference/ssr/models/__init__.py:8-11."""
import importlib
import os

_here = os.path.dirname(os.path.abspath(__file__))
_model_modules = [importlib.import_module(f"{__name__}.{f[:-3]}") for f in sorted(os.listdir(_here))
                  if f.endswith("_model.py")]
