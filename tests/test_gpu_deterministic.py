"""GPU: the deterministic mode (StepConfig.deterministic / SSR_DETERMINISTIC=1 / option-file key `deterministic: true`).

The reference's fp32 CPU path is bit-identical from run to run.  The default HIP step is not: four
reductions used to meet in arrival order -
  1. weight gradients of layers that are split over pixel ranges (the 128^2 / 64^2 layers): fp32 atomics into one buffer;
  2. the bias-gradient partial sums of the bf16 weight-gradient kernels: LDS float atomics;
  3. <dW_sn, W> of the spectral-norm backward: one fp32 atomic per block;
  4. the logged loss scalars: one fp32 atomic per block.
Round 4: 2 and 3 are fixed-order sums in every mode (xor-shuffle tree + waves in turn; per-block slots added in index order);
1 and 4 are fixed-order behind the switch (per-split partial gradients + ssr_wgrad_reduce; per-block loss slots).

Done = two independent runs of 3 optimize_parameters() steps end with bit-identical generator / discriminator parameters,
spectral-norm vectors, Adam moments, EMA and logged losses, in the bf16 mode (fused dense-block kernels, bf16 wgrad kernels)
and in fp32x3; and the deterministic step computes the same thing as the default one (to rounding).  That two data-parallel
replicas stay bit-identical is asserted by tests/test_dp_gpu.py::test_dp_world2_shared_gpu_matches_full_batch_oracle."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

G_KW = dict(num_in_ch=24, num_out_ch=3, scale=4, num_feat=64, num_block=2, num_grow_ch=32)
D_KW = dict(num_in_ch=3, num_feat=64, skip_connection=True)
B = 4


def _run(mode, deterministic, use_graph, steps=3):
    from oracle import esrgan_oracle as O
    from satlas_super_resolution_amd.train_step import ESRGANTrainStep, StepConfig
    ts = ESRGANTrainStep(G_KW, D_KW, B, 32, 32, mode, StepConfig(deterministic=deterministic), use_graph=use_graph)
    ts.load_state(O.generator_init(seed=21, **G_KW), O.discriminator_init(3, 64, seed=22))
    g = torch.Generator().manual_seed(23)
    logs = []
    for it in range(1, steps + 1):
        lr, gt = torch.rand(B, 24, 32, 32, generator=g), torch.rand(B, 3, 128, 128, generator=g)
        ts.feed_data(lr.cuda(), gt.cuda())
        ts.step(it)
        logs.append(dict(ts.log()))
    torch.cuda.synchronize()
    state = {"g": ts.g_store.data.cpu().clone(), "d": ts.d_store.data.cpu().clone(), "ema": ts.opt_g.ema.cpu().clone(),
             "g_m": ts.opt_g.exp_avg.cpu().clone(), "g_v": ts.opt_g.exp_avg_sq.cpu().clone(),
             "d_m": ts.opt_d.exp_avg.cpu().clone(), "d_v": ts.opt_d.exp_avg_sq.cpu().clone(),
             "g_grad": ts.g_store.grad.cpu().clone(), "d_grad": ts.d_store.grad.cpu().clone()}
    for k, v in ts.d_store.u.items():
        state["u." + k] = v.cpu().clone()
    n_partial = sum(len(b._partials) for b in getattr(ts.g_plan, "_wg_batches", []) if b is not None)
    return state, logs, n_partial


@pytest.mark.parametrize("mode,use_graph", [("bf16", False), ("bf16", True), ("fp32x3", False)])
def test_two_runs_of_three_steps_are_bit_identical(mode, use_graph):
    a, la, npart = _run(mode, True, use_graph)
    b, lb, _ = _run(mode, True, use_graph)
    assert npart > 0, "the test shape must contain layers that are split over pixel ranges (else nothing is exercised)"
    for k in a:
        assert torch.equal(a[k], b[k]), f"{mode}: '{k}' differs between two deterministic runs (max {float((a[k] - b[k]).abs().max()):.3e})"
    assert la == lb, (la, lb)
    assert all(v == v for lg in la for v in lg.values())


def test_deterministic_step_equals_the_default_step_to_rounding():
    a, la, _ = _run("bf16", True, False, steps=1)
    b, lb, npart = _run("bf16", False, False, steps=1)
    assert npart == 0
    # one step: the gradients are the same sums in another order
    assert rel_err(a["g_grad"], b["g_grad"]) < 1e-5 and rel_err(a["d_grad"], b["d_grad"]) < 1e-5
    for k in la[0]:
        assert abs(la[0][k] - lb[0][k]) <= 1e-5 * max(1.0, abs(lb[0][k])), (k, la[0][k], lb[0][k])
