"""BasicSR-compatible registries.

The reference registers its plugins with `@ARCH_REGISTRY.register()` / `@MODEL_REGISTRY.register()`
from `basicsr.utils.registry` (/root/reference/ssr/archs/rrdbnet_arch.py:10,71;
discriminator_arch.py:9,11; ssr/models/ssr_esrgan_model.py:13,18) and BasicSR resolves
`network_g.type` / `model_type` from the YAML through `REGISTRY.get(name)`.  When BasicSR is installed
we register into ITS registries (so `basicsr.archs.build_network` finds our classes under the same
names); otherwise an API-identical local registry is used."""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj, suffix=None):
        if isinstance(suffix, str):
            name = name + "_" + suffix
        assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
        self._obj_map[name] = obj

    def register(self, obj=None, suffix=None):
        if obj is None:
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class, suffix)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj, suffix)
        return obj

    def get(self, name, suffix="basicsr"):
        ret = self._obj_map.get(name)
        if ret is None:
            ret = self._obj_map.get(name + "_" + suffix)
        if ret is None:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return ret

    def __contains__(self, name):
        return name in self._obj_map

    def keys(self):
        return self._obj_map.keys()


try:  # pragma: no cover - BasicSR is not installed in the build image
    from basicsr.utils.registry import ARCH_REGISTRY, DATASET_REGISTRY, MODEL_REGISTRY  # type: ignore
    if not hasattr(ARCH_REGISTRY, "register"):
        raise ImportError
except Exception:
    ARCH_REGISTRY = Registry("arch")
    MODEL_REGISTRY = Registry("model")
    DATASET_REGISTRY = Registry("dataset")


def build_network(opt: dict):
    """basicsr.archs.build_network: `type` selects the class, the rest are ctor kwargs."""
    opt = dict(opt)
    return ARCH_REGISTRY.get(opt.pop("type"))(**opt)


def build_model(opt: dict):
    """basicsr.models.build_model (train.py:62)."""
    return MODEL_REGISTRY.get(opt["model_type"])(opt)


def build_dataset(dataset_opt: dict):
    """basicsr.data.build_dataset: `type` selects the class, the whole dict is its `opt`."""
    return DATASET_REGISTRY.get(dataset_opt["type"])(dataset_opt)
