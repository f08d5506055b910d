"""TEST INFRASTRUCTURE — golden vectors of the UNMODIFIED reference classes at the BENCHMARKED architecture size
(SSR_RRDBNet nf=64 / gc=32 / nb=23 with 24 and 96 input channels on 32x32 tiles; SSR_UNetDiscriminatorSN nf=64 on 128x128 with 3
and 27 input channels).  A 67 MB state_dict does not belong in git, so the parameters are NOT the reference's own random draw:
they come from `oracle.esrgan_oracle.generator_init / discriminator_init(seed)` and are loaded into the reference modules with
`load_state_dict(strict=True)`; a test rebuilds them from the seed.  Stored: the seeded input, the reference module's output,
input gradient and a handful of parameter gradients (early / middle / late layers); inputs are regenerated from the seed.  Run in the build container:
    python oracle/make_golden_fullsize.py        ->  tests/golden/full_g24.pt, full_g96.pt, full_d3.pt, full_d27.pt"""
import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import esrgan_oracle as O  # noqa: E402
from oracle.ref_shim import load_reference_archs  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
G_KEEP = ["conv_first.weight", "conv_first.bias", "body.0.rdb1.conv1.weight", "body.11.rdb2.conv3.weight", "body.22.rdb3.conv5.bias",
          "body.22.rdb3.conv5.weight", "conv_body.weight", "conv_up2.weight", "conv_hr.bias", "conv_last.weight"]
D_KEEP = ["conv0.weight", "conv0.bias", "conv1.weight_orig", "conv7.weight_orig", "conv8.weight_orig", "conv9.weight", "conv9.bias"]


def biased(sd, seed):
    g = torch.Generator().manual_seed(seed)
    for k in list(sd):
        if k.endswith(".bias"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
    return sd


def gen_g(name, c_in, seed):
    G, _, _ = load_reference_archs()
    kw = dict(num_in_ch=c_in, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
    sd = biased(O.generator_init(seed=seed, **kw), seed + 1)
    net = G(**kw).train()
    net.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.rand(2, c_in, 32, 32, generator=g).requires_grad_(True)
    y = net(x)
    r = torch.randn(y.shape, generator=g)
    (y * r).sum().backward()
    grads = OrderedDict((n, p.grad.detach().clone()) for n, p in net.named_parameters() if n in G_KEEP)
    assert len(grads) == len(G_KEEP)
    # x and r are regenerated from the seed by the tests (torch.rand / randn with a seeded CPU Generator)
    torch.save({"kwargs": kw, "seed": seed, "x_check": x.detach()[0, :, 0, 0].clone(), "y": y.detach(), "dx": x.grad.detach(), "grads": grads},
               os.path.join(OUT, name + ".pt"))
    print(name, tuple(y.shape), float(y.abs().max()))


def gen_d(name, c_d, seed):
    _, D, _ = load_reference_archs()
    sd = O.discriminator_init(c_d, 64, seed=seed)
    net = D(num_in_ch=c_d, num_feat=64, skip_connection=True).train()
    net.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.rand(1, c_d, 128, 128, generator=g).requires_grad_(True)
    y = net(x)                                  # train mode: one power iteration
    r = torch.randn(y.shape, generator=g)
    (y * r).sum().backward()
    grads = OrderedDict((n, p.grad.detach().clone()) for n, p in net.named_parameters() if n in D_KEEP)
    assert len(grads) == len(D_KEEP)
    uv = OrderedDict((k, v.detach().clone()) for k, v in net.state_dict().items() if k.endswith(("_u", "_v")))
    dx = x.grad.detach()
    torch.save({"c_d": c_d, "seed": seed, "x_check": x.detach()[0, :, 0, 0].clone(), "y": y.detach(),
                "dx_first3": dx[:, :3].clone(), "dx_last3": dx[:, -3:].clone(), "grads": grads, "uv_after": uv},
               os.path.join(OUT, name + ".pt"))
    print(name, tuple(y.shape), float(y.abs().max()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    gen_g("full_g24", 24, 101)
    gen_g("full_g96", 96, 111)
    gen_d("full_d3", 3, 121)
    gen_d("full_d27", 27, 131)
