"""Testing entry point with the control flow of /root/reference/ssr/test.py:14-46 (no BasicSR needed): every dataset under
`test_datasets` is built, the model is constructed with is_train=False (generator only, weights from `path.pretrain_network_g` /
`param_key_g`) and `model.validation` runs the `test.metrics` of the option file and writes the images.

    python -m satlas_super_resolution_amd.test -opt <yml>"""
from __future__ import annotations

import argparse
from typing import Dict

import torch


def test_pipeline(opt: Dict, log=print) -> Dict[str, Dict[str, float]]:
    from . import data as _data, models as _models  # noqa: F401  (register the plugins)
    from .registry import build_dataset, build_model
    opt = dict(opt, is_train=False, dist=False)
    loaders = []
    for _, dopt in sorted(opt["test_datasets"].items()):
        dset = build_dataset(dict(dopt, phase=dopt.get("phase", "test"), scale=dopt.get("scale", opt.get("scale", 4))))
        log(f"Number of test images in {dopt['name']}: {len(dset)}")
        loaders.append(torch.utils.data.DataLoader(dset, batch_size=1, shuffle=False, num_workers=0))
    model = build_model(opt)
    results = {}
    for loader in loaders:
        name = loader.dataset.opt["name"]
        log(f"Testing {name}...")
        model.validation(loader, current_iter=opt.get("name", "test"), tb_logger=None, save_img=opt.get("test", {}).get("save_img", False))
        results[name] = dict(model.metric_results)
    return results


def main():
    import yaml
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", type=str, required=True, help="Path to option YAML file.")
    args = ap.parse_args()
    with open(args.opt) as f:
        opt = yaml.safe_load(f)
    print(test_pipeline(opt))


if __name__ == "__main__":
    main()
