"""Fused clip + Adam (train.ClipAdam -> csrc/adam.hip) against torch.optim.Adam + clip_grad_norm_ - the pair the reference
gets from Lightning (scripts/train_poseestimator.py:147-167, :442-445): checkpoint resume, per-parameter step counts
(a parameter without a gradient is not stepped), data-parallel gradient scale, stable device tables."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(1024, 1024), (50, 1024), (7,), (3, 3, 5), (4097,), (1,)]


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in SHAPES]


def _grads(step, skip=()):
    g = torch.Generator().manual_seed(100 + step)
    return [None if i in skip else (torch.randn(s, generator=g) * (10.0 if step % 2 else 0.01)).to(DEV) for i, s in enumerate(SHAPES)]


def _groups(ps):
    return [{"params": ps[:3], "lr": 1e-3}, {"params": ps[3:5], "lr": 1e-4}, {"params": ps[5:], "lr": 1e-5, "weight_decay": 0.01}]


def _reference_run(steps, skips):
    ps = _params()
    opt = torch.optim.Adam(_groups(ps), lr=1e-3)
    for s in range(steps):
        for p, g in zip(ps, _grads(s, skips.get(s, ()))):
            p.grad = g
        torch.nn.utils.clip_grad_norm_(ps, 1.0)
        opt.step()
    return ps, opt


def test_matches_torch_adam_with_skipped_parameters_and_resume():
    from trackertraincode.train import ClipAdam

    skips = {1: (2, 4), 2: (2,)}  # parameters 2 and 4 have no gradient in some steps: torch does not advance their step count
    ref_ps, ref_opt = _reference_run(5, skips)

    ps = _params()
    opt = ClipAdam(_groups(ps), lr=1e-3, max_norm=1.0)
    for s in range(3):
        for p, g in zip(ps, _grads(s, skips.get(s, ()))):
            p.grad = g
        opt.step()
    # ---- checkpoint, resume into a fresh optimiser over copies of the parameters
    sd = copy.deepcopy(opt.state_dict())
    assert float(sd["state"][2]["step"]) == 1.0 and float(sd["state"][0]["step"]) == 3.0  # per-parameter counts
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt2 = ClipAdam(_groups(ps2), lr=1e-3, max_norm=1.0)
    opt2.load_state_dict(sd)
    assert opt2._t == 3
    for s in range(3, 5):
        for p, g in zip(ps2, _grads(s, skips.get(s, ()))):
            p.grad = g
        opt2.step()
    torch.cuda.synchronize()
    for a, b in zip(ps2, ref_ps):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=2e-7)
    for a, b in zip(ps2, ref_ps):
        sa, sb = opt2.state[a], ref_opt.state[b]
        assert float(sa["step"]) == float(sb["step"])
        np.testing.assert_allclose(sa["exp_avg"].cpu().numpy(), sb["exp_avg"].cpu().numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(sa["exp_avg_sq"].cpu().numpy(), sb["exp_avg_sq"].cpu().numpy(), rtol=5e-5, atol=1e-12)  # g^2 after the clip coefficient: twice its fp32 rounding
    # a load AFTER steps were taken must drop every cached device address (the old moments are orphaned otherwise)
    opt2.load_state_dict(sd)
    assert opt2._tables is None and opt2._t == 3


def test_grad_scale_equals_prescaled_gradients():
    """grad_scale = 1/world with summed gradients in memory == plain step on the averaged gradients."""
    from trackertraincode.train import ClipAdam

    world = 8
    a, b = _params(), _params()
    oa, ob = ClipAdam(_groups(a), max_norm=1.0), ClipAdam(_groups(b), max_norm=1.0)
    ob.grad_scale = 1.0 / world
    for s in range(3):
        for p, q, g in zip(a, b, _grads(s)):
            p.grad, q.grad = g, g * world
        oa.step()
        ob.step()
        np.testing.assert_allclose(ob.last_grad_norm.item(), oa.last_grad_norm.item(), rtol=1e-6)
    for p, q in zip(a, b):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_pointer_table_follows_new_gradient_addresses():
    """Gradients that live at new addresses every step (fresh tensors, the old ones kept alive) must be the ones read."""
    from trackertraincode.train import ClipAdam

    ref_ps, _ = _reference_run(4, {})
    ps = _params()
    opt = ClipAdam(_groups(ps), max_norm=1.0)
    keep = []
    for s in range(4):
        gs = _grads(s)
        keep.append(gs)  # nothing is freed: the allocator cannot hand the same addresses out again
        for p, g in zip(ps, gs):
            p.grad = g
        opt.step()  # no synchronisation between steps: the host runs ahead of the GPU
    torch.cuda.synchronize()
    for a, b in zip(ps, ref_ps):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=2e-7)
