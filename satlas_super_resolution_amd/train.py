"""A BasicSR-free training loop over the plugins of this package, following the control flow of /root/reference/ssr/train.py:52-140
(build loaders -> build_model -> [resume] -> per iteration: update_learning_rate, feed_data, optimize_parameters, log, save,
validation) with the same option-file keys.  The reference's own train.py drives these plugins unchanged when BasicSR is installed
(INTEGRATION.md); this loop is for hosts without it:

    python -m satlas_super_resolution_amd.train -opt ssr/options/esrgan_s2naip_urban.yml [--launcher pytorch]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m satlas_super_resolution_amd.train -opt ... --launcher pytorch

Not reproduced: BasicSR's loggers (tensorboard / wandb), experiment-directory bookkeeping and the CUDA prefetcher (batches are
uint8 and uploaded asynchronously from pinned memory by the DataLoader)."""
from __future__ import annotations

import argparse
import json
import os
import random
import time
from typing import Dict, Optional

import torch


def train(opt: Dict, max_iters: Optional[int] = None, resume_state: Optional[Dict] = None, log=print) -> Dict:
    from . import data as _data, models as _models  # noqa: F401  (register the plugins)
    from .data.s2naip_dataset import build_train_loader
    from .registry import build_dataset, build_model
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    seed = opt.get("manual_seed")
    if seed is not None:      # options.py:79-82: every rank its own stream
        random.seed(seed + rank)
        torch.manual_seed(seed + rank)
    dsets = opt["datasets"]
    train_opt = dict(dsets["train"], phase="train", scale=opt.get("scale", 4))
    train_set = build_dataset(train_opt)
    loader = build_train_loader(train_set, train_opt, rank=rank, world=world, seed=seed)
    val_loaders = []
    for phase, dopt in dsets.items():
        if phase.split("_")[0] == "val":
            vset = build_dataset(dict(dopt, phase="val", scale=opt.get("scale", 4)))
            val_loaders.append(torch.utils.data.DataLoader(vset, batch_size=1, shuffle=False, num_workers=0))
    total_iters = int(opt["train"]["total_iter"]) if max_iters is None else max_iters
    model = build_model(opt)
    current_iter, epoch = 0, 0
    if resume_state:
        model.resume_training(resume_state)
        current_iter, epoch = resume_state["iter"], resume_state["epoch"]
        log(f"Resuming training from epoch: {epoch}, iter: {current_iter}.")
    logger_opt = opt.get("logger", {})
    t0, seen = time.time(), 0
    while current_iter < total_iters:
        if hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)
        for batch in loader:
            if current_iter >= total_iters:      # (the reference's loop counts one past the end before it breaks: train.py:104-108;
                break                            #  the returned iteration count is the number of optimizer steps taken)
            current_iter += 1
            model.update_learning_rate(current_iter, warmup_iter=opt["train"].get("warmup_iter", -1))
            model.feed_data(batch)
            model.optimize_parameters(current_iter)
            seen += batch["lr"].shape[0]
            if current_iter % int(logger_opt.get("print_freq", 100)) == 0:
                # get_current_log() reduces the loss scalars over the ranks (a collective): EVERY rank calls it, as the
                # reference's loop does (ssr/train.py:111-117 -> reduce_loss_dict); only rank 0 prints
                cur_log = model.get_current_log()
                if rank == 0:
                    msg = {"epoch": epoch, "iter": current_iter, "lrs": model.get_current_learning_rate(),
                           "img_per_s": round(world * seen / (time.time() - t0), 1), **cur_log}
                    log(json.dumps(msg))
            if current_iter % int(float(logger_opt.get("save_checkpoint_freq", 5e3))) == 0:
                model.save(epoch, current_iter)
            if opt.get("val") is not None and current_iter % int(float(opt["val"]["val_freq"])) == 0:
                for vl in val_loaders:
                    model.validation(vl, current_iter, None, opt["val"].get("save_img", False))
                if rank == 0 and val_loaders:
                    log(json.dumps({"iter": current_iter, "validation": model.metric_results}))
        epoch += 1
    model.save(epoch=-1, current_iter=-1)      # "latest"
    return {"iters": current_iter, "epochs": epoch, "log": model.get_current_log(), "metrics": dict(model.metric_results)}


def resolve_resume(opt: Dict, auto_resume: bool = False, log=print) -> Optional[Dict]:
    """BasicSR's load_resume_state + check_resume (train.py:64-65 of the reference calls them): `--auto_resume` takes the newest
    `training_states/<iter>.state` and OVERRIDES `path.resume_state`; a resume ALWAYS redirects `pretrain_network_{g,d}` to
    `models/net_{g,d}_<iter>.pth` (unless the net is listed in `path.ignore_resume_networks`), and a missing file is an error, not a
    silent fall-back to the pretrain weights: optimizer moments, EMA and iteration counters of the state file belong to those
    weights and to no others; every `path.param_key_*` that says 'params_ema' is reset to 'params'."""
    path = opt["path"]
    state_file = None
    if auto_resume and os.path.isdir(path["training_states"]):
        states = [f for f in os.listdir(path["training_states"]) if f.endswith(".state") and f[:-6].isdigit()]
        if states:
            state_file = os.path.join(path["training_states"], f"{max(int(f[:-6]) for f in states)}.state")
    if state_file is None and path.get("resume_state"):
        state_file = path["resume_state"]
    if state_file is None:
        return None
    resume = torch.load(state_file, map_location="cpu", weights_only=False)
    it = resume["iter"]
    ignore = path.get("ignore_resume_networks") or []
    for net in ("g", "d"):
        if f"network_{net}" in ignore or net in ignore:
            log(f"resume: keeping pretrain_network_{net} (ignore_resume_networks)")
            continue
        cand = os.path.join(path["models"], f"net_{net}_{it}.pth")
        if not os.path.exists(cand):
            raise FileNotFoundError(f"resume state {state_file} is at iteration {it} but {cand} does not exist "
                                    f"(list 'network_{net}' in path.ignore_resume_networks to keep pretrain_network_{net})")
        if path.get(f"pretrain_network_{net}") not in (None, cand):
            log(f"resume: pretrain_network_{net} is redirected to {cand}")
        path[f"pretrain_network_{net}"] = cand
    # check_resume's last step: a checkpoint written by save() holds the TRAINED weights under 'params' and the EMA under
    # 'params_ema'; the option files read `param_key_g: params_ema` for fine-tuning / inference, and on a resume that would load
    # the EMA into the trainable generator while the restored Adam moments and counters belong to 'params'
    for k in [k for k in path if k.startswith("param_key")]:
        if path[k] == "params_ema":
            path[k] = "params"
            log(f"resume: {k} is reset from 'params_ema' to 'params' (the optimizer state belongs to the raw weights)")
    return resume


def main():
    import yaml
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", type=str, required=True, help="Path to option YAML file.")
    ap.add_argument("--launcher", choices=["none", "pytorch"], default="none")
    ap.add_argument("--auto_resume", action="store_true")
    ap.add_argument("--max-iters", type=int, default=None)
    args = ap.parse_args()
    with open(args.opt) as f:
        opt = yaml.safe_load(f)
    opt["is_train"] = True
    opt["dist"] = args.launcher == "pytorch" and int(os.environ.get("WORLD_SIZE", "1")) > 1
    name = opt.get("name", "run")
    root = os.path.join("experiments", name)
    opt.setdefault("path", {})
    opt["path"] = {k: v for k, v in (opt["path"] or {}).items()}
    opt["path"].setdefault("models", os.path.join(root, "models"))
    opt["path"].setdefault("training_states", os.path.join(root, "training_states"))
    opt["path"].setdefault("visualization", os.path.join(root, "visualization"))
    resume = resolve_resume(opt, args.auto_resume)
    train(opt, max_iters=args.max_iters, resume_state=resume)


if __name__ == "__main__":
    main()
