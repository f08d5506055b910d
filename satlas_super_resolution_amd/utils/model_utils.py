"""`build_network` of the inference scripts (/root/reference/ssr/utils/model_utils.py:5-18): constructs the generator
directly from the YAML keys (`num_in_ch = n_lr_images * 3`), no registry lookup."""


def build_network(opt: dict):
    n_lr_images = opt["n_lr_images"]
    net = opt["network_g"]
    if net["type"] != "SSR_RRDBNet":
        raise NotImplementedError(f"network_g.type {net['type']!r}: only SSR_RRDBNet is on the MI355X hot path")
    from ..archs.rrdbnet_arch import SSR_RRDBNet
    kw = {k: v for k, v in net.items() if k not in ("type", "num_in_ch")}
    # model_utils.py:13-18: num_in_ch is derived from n_lr_images (use_3d / other archs are outside the hot path)
    return SSR_RRDBNet(num_in_ch=n_lr_images * 3, **kw)
