"""CPU, world_size 2 / 4 / 8, gloo: the data-parallel exchange (satlas_super_resolution_amd/dp.py; the reference launches 8 ranks,
/root/reference/README.md:159).

Checks (a) chunked async all-reduce of a flat arena, (b) the DP identity the design relies on:
per-rank half-batch gradients summed and scaled by 1/world equal the full-batch gradients of the
mean-reduced losses — for G, and for D with ONE exchange after both backward passes (instead of the
reference DDP's two), and (c) that spectral-norm u/v stay identical across ranks without a broadcast.
The compute is the CPU oracle (test infrastructure)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from satlas_super_resolution_amd.dp import init_distributed
    from oracle import esrgan_oracle as O
    torch.set_num_threads(1 if world > 4 else 2)
    ctx = init_distributed(backend="gloo")
    assert ctx.world == world and ctx.rank == rank and ctx.active
    # (a) chunked all-reduce over a flat arena
    ctx.chunk_elems = 1000
    flat = torch.arange(4321, dtype=torch.float32) * (rank + 1)
    ctx.all_reduce_async(flat)
    ctx.wait()
    assert torch.equal(flat, torch.arange(4321, dtype=torch.float32) * (world * (world + 1) // 2))
    assert abs(ctx.grad_scale - 1.0 / world) < 1e-12
    # (a2) the segmented generator exchange of train_step.step(): the gradient arena is exchanged slice by slice, LAST slice first
    # (engine.GeneratorPlan.bwd_segments: contiguous slices in backward order), each slice chunked on its own, each with its own
    # handle; slice and chunk boundaries do not coincide (chunk_elems = 1000 against slices of 1501 / 1700 / 1120 elements)
    arena = (torch.arange(4321, dtype=torch.float32) % 97) * (rank + 1)
    bounds = [(2621, 4321), (1120, 2621), (0, 1120)]
    assert sum(hi - lo for lo, hi in bounds) == arena.numel()
    handles = [ctx.all_reduce_async(arena[lo:hi]) for lo, hi in bounds]
    assert [len(h) for h in handles] == [2, 2, 2]          # ceil(1700 / 1000), ceil(1501 / 1000), ceil(1120 / 1000) collectives
    for h in handles:
        ctx.wait(h)
    assert not ctx._pending
    assert torch.equal(arena, (torch.arange(4321, dtype=torch.float32) % 97) * (world * (world + 1) // 2))
    # (a3) SSR_DP_ALGO=rsag: the same exchange as reduce-scatter + all-gather over the (padded) arena - lengths that are / are not a
    # multiple of the world size, shorter than the world size, and slices of a larger arena (the segmented generator exchange)
    from satlas_super_resolution_amd.dp import DPContext
    c2 = DPContext(ctx.group, rank, world, algo="rsag")
    for n in (4321, 4096, 5, 8 * 17):
        flat = (torch.arange(n, dtype=torch.float32) % 89) * (rank + 1)
        c2.wait(c2.all_reduce_async(flat))
        assert torch.equal(flat, (torch.arange(n, dtype=torch.float32) % 89) * (world * (world + 1) // 2)), n
    arena = (torch.arange(4321, dtype=torch.float32) % 97) * (rank + 1)
    for lo, hi in bounds:
        c2.wait(c2.all_reduce_async(arena[lo:hi]))
    assert torch.equal(arena, (torch.arange(4321, dtype=torch.float32) % 97) * (world * (world + 1) // 2))
    # (b) DP identity on a small ESRGAN step
    g_kw = dict(num_in_ch=6, num_out_ch=3, scale=4, num_feat=16, num_block=1, num_grow_ch=8)
    g0 = O.generator_init(seed=5, **g_kw)
    d0 = O.discriminator_init(3, 8, seed=6)
    torch.manual_seed(7)
    lr, gt = torch.rand(world, 6, 8, 8), torch.rand(world, 3, 32, 32)      # global batch = world x 1 (weak scaling: per-rank batch fixed)
    full = O.ESRGANOracle(g0, d0, O.StepConfig())
    full.step(lr, gt, 1)
    part = O.ESRGANOracle(g0, d0, O.StepConfig())
    part.step(lr[rank:rank + 1], gt[rank:rank + 1], 1)
    keys_g, keys_d = list(part.g_grads), list(part.d_grads)
    flat_g = torch.cat([part.g_grads[k].reshape(-1) for k in keys_g])
    flat_d = torch.cat([part.d_grads[k].reshape(-1) for k in keys_d])     # real+fake already summed: ONE exchange
    hg = ctx.all_reduce_async(flat_g)
    hd = ctx.all_reduce_async(flat_d)
    ctx.wait(hg)                                                         # per-exchange handles, as train_step.step() uses them
    assert len(ctx._pending) == len(hd)
    ctx.wait(hd)
    assert not ctx._pending
    flat_g *= ctx.grad_scale
    flat_d *= ctx.grad_scale
    ref_g = torch.cat([full.g_grads[k].reshape(-1) for k in keys_g])
    ref_d = torch.cat([full.d_grads[k].reshape(-1) for k in keys_d])
    eg = float((flat_g - ref_g).abs().max() / ref_g.abs().max())
    ed = float((flat_d - ref_d).abs().max() / ref_d.abs().max())
    # (c) u/v identical across ranks without broadcast (same weights -> same power iteration)
    u = part.d["conv3.weight_u"].clone()
    gathered = [torch.zeros_like(u) for _ in range(world)]
    dist.all_gather(gathered, u)
    same_uv = all(torch.equal(gathered[0], t) for t in gathered)
    losses = ctx.reduce_scalars(torch.tensor([float(rank + 1), 2.0]))
    q.put((rank, eg, ed, same_uv, losses.tolist()))
    ctx.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dp_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=480) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, eg, ed, same_uv, losses in res:
        assert eg < 1e-4, ("G grads", rank, eg)
        assert ed < 1e-4, ("D grads", rank, ed)
        assert same_uv
        assert losses == [(world + 1) / 2, 2.0]
