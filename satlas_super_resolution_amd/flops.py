"""Algorithmic FLOP model of the ESRGAN G+D train step (SURVEY.md section 8d / BASELINE.md section 3) and random initial states with the
reference's init distributions - product accounting used by bench.py and the tools, derived from the layer lists the launch plans
are built from (engine.generator_specs / engine.discriminator_specs).

A convolution's MACs per image = Cout * Cin * k * k * (output pixels).  Output grids: the generator's conv_first / body / conv_body run
on the input grid, conv_up{i} on 2^i times it, conv_hr / conv_last on the output grid (/root/reference/ssr/archs/rrdbnet_arch.py:116-137);
the discriminator's conv0 / conv6-9 on the input grid, conv1-3 on 1/2, 1/4, 1/8 of it, conv4 / conv5 on 1/4, 1/2 (the U-Net of
/root/reference/ssr/archs/discriminator_arch.py:42-71)."""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import engine


def generator_conv_macs(num_in_ch: int, H: int = 32, W: int = 32, nf: int = 64, nb: int = 23, gc: int = 32, num_out_ch: int = 3,
                        scale: int = 4) -> Dict[str, int]:
    """MACs per image of every generator convolution, keyed by layer name; the dense blocks summed under "body" """
    out: Dict[str, int] = {"body": 0}
    px = H * W if scale == 4 else (H // (2 if scale == 2 else 4 if scale == 1 else 1)) * (W // (2 if scale == 2 else 4 if scale == 1 else 1))
    n_up = sum(1 for s in engine.generator_specs(num_in_ch, num_out_ch, scale, nf, nb, gc) if s.name.startswith("conv_up"))
    for s in engine.generator_specs(num_in_ch, num_out_ch, scale, nf, nb, gc):
        if s.name.startswith("conv_up"):
            grid = px << (2 * int(s.name[len("conv_up"):]))
        elif s.name in ("conv_hr", "conv_last"):
            grid = px << (2 * n_up)
        else:
            grid = px
        macs = s.cout * s.cin * s.k * s.k * grid
        if s.name.startswith("body."):
            out["body"] += macs
        else:
            out[s.name] = macs
    return out


_D_GRID_SHIFT = {"conv0": 0, "conv1": 1, "conv2": 2, "conv3": 3, "conv4": 2, "conv5": 1, "conv6": 0, "conv7": 0, "conv8": 0, "conv9": 0}


def discriminator_conv_macs(num_in_ch: int, H: int = 128, W: int = 128, nf: int = 64) -> Dict[str, int]:
    return {s.name: s.cout * s.cin * s.k * s.k * ((H >> _D_GRID_SHIFT[s.name]) * (W >> _D_GRID_SHIFT[s.name]))
            for s in engine.discriminator_specs(num_in_ch, nf)}


def step_gflop_per_image(c_in: int, c_d: int, nb: int = 23) -> float:
    """One optimize_parameters() per image, in GFLOP (2 FLOP per MAC): forward + input gradient + weight gradient of every conv of G,
    minus conv_first's input gradient (autograd skips it); the discriminator 3 forwards + 3 input-gradient passes (minus conv0's in the
    two D phases) + 2 weight-gradient passes = 8 D_fwd - 2 MAC(conv0)  (/root/reference/ssr/models/ssr_esrgan_model.py:140,181,192,217-227)."""
    g = generator_conv_macs(c_in, nb=nb)
    d = discriminator_conv_macs(c_d)
    macs = 3 * sum(g.values()) - g["conv_first"] + 8 * sum(d.values()) - 2 * d["conv0"]
    return 2.0 * macs / 1e9


def conv_launch_flops(d) -> float:
    """algorithmic FLOPs of one ssr_conv2d launch described by an ssr_conv_desc: 2 * grid * Cout * taps * contracted channels"""
    return 2.0 * d.N * d.Gh * d.Gw * d.Cout * d.KH * d.KW * (d.Cin + d.Cin2)


RDB_MACS_PER_PIXEL = 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64)      # one dense block at nf = 64, gc = 32


def random_state(specs: List[engine.ConvSpec], seed: Optional[int] = None, rdb_prefix: str = "body.") -> "OrderedDict[str, torch.Tensor]":
    """A state_dict in the reference's key layout with the reference's init DISTRIBUTIONS: torch's Conv2d default (uniform, bound
    1/sqrt(fan_in)) everywhere except the dense-block convs, which get kaiming_normal * 0.1 and zero bias
    (/root/reference/ssr/archs/arch_util.py:600-628 via rrdbnet_arch.py:35); spectral-norm layers carry weight_orig and unit-norm
    gaussian u / v as torch.nn.utils.spectral_norm creates them (discriminator_arch.py:30-39).  Random weights of the named
    architecture for benchmarks and tests - not the reference's RNG stream (the golden fixtures hold its actual tensors)."""
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for s in specs:
        fan_in = s.cin * s.k * s.k
        bound = 1.0 / math.sqrt(fan_in)
        if s.name.startswith(rdb_prefix):
            w = torch.randn(s.cout, s.cin, s.k, s.k, generator=g) * (math.sqrt(2.0 / fan_in) * 0.1)
            b = torch.zeros(s.cout)
        else:
            w = (torch.rand(s.cout, s.cin, s.k, s.k, generator=g) * 2 - 1) * bound
            b = (torch.rand(s.cout, generator=g) * 2 - 1) * bound if s.bias else None
        if s.sn:
            sd[s.name + ".weight_orig"] = w
            sd[s.name + ".weight_u"] = F.normalize(torch.randn(s.cout, generator=g), dim=0, eps=1e-12)
            sd[s.name + ".weight_v"] = F.normalize(torch.randn(fan_in, generator=g), dim=0, eps=1e-12)
        else:
            sd[s.name + ".weight"] = w
        if s.bias:
            sd[s.name + ".bias"] = b
    return sd


def generator_random_state(seed: Optional[int] = None, **g_kw):
    return random_state(engine.generator_specs(**g_kw), seed)


def discriminator_random_state(num_in_ch: int, num_feat: int = 64, seed: Optional[int] = None):
    return random_state(engine.discriminator_specs(num_in_ch, num_feat), seed, rdb_prefix="\0")
