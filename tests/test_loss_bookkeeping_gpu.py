"""The single-launch loss bookkeeping of a training step (ttk_multi_copy, ttk_weighted_sum_*, ttk_loss_mse_cols_*, and
the autograd nodes around them) against the reference's plain formulas (train.py:372-439, losses.py:66-85) in torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_multi_copy_segments_and_zero_fill():
    import trackertraincode._hip as H
    L = H.lib()
    g = torch.Generator().manual_seed(0)
    sizes = [1, 7, 0, 513, 4096, 33] * 7  # 42 segments: two launches
    srcs = [torch.randn(n, generator=g).cuda() if i % 5 else None for i, n in enumerate(sizes)]
    dsts = [torch.full((n,), float("nan"), device="cuda") for n in sizes]
    L.multi_copy(srcs, dsts)
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, torch.zeros_like(d) if s is None else s)
    with pytest.raises(RuntimeError, match="multi_copy"):
        L.multi_copy([torch.zeros(3, device="cuda")], [torch.zeros(4, device="cuda")])


def test_mse_cols_matches_sliced_formula():
    from trackertraincode.neuralnets import _hipops
    g = torch.Generator().manual_seed(1)
    p0, t = torch.randn(300, 3, generator=g).cuda(), torch.randn(300, 3, generator=g).cuda()
    gv = torch.randn(300, generator=g).cuda()
    for c0, nc in ((0, 2), (2, 1), (0, 3)):
        p = p0.clone().requires_grad_(True)
        v = _hipops.mse_cols(p, t, c0, nc)
        v.backward(gv)
        q = p0.clone().requires_grad_(True)
        ref = ((q[:, c0:c0 + nc] - t[:, c0:c0 + nc]) ** 2).mean(-1)
        ref.backward(gv)
        torch.testing.assert_close(v, ref, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-6, atol=1e-7)


def test_split_rows_backward_assembles_gradients():
    from trackertraincode.neuralnets import _hipops
    g = torch.Generator().manual_seed(2)
    a0, b0, c0 = torch.randn(10, 3, generator=g).cuda(), torch.randn(10, 68, 3, generator=g).cuda(), torch.randn(10, generator=g).cuda()
    sizes = (4, 0, 6)

    def run(split):
        a, b, c = (x.clone().requires_grad_(True) for x in (a0, b0, c0))
        if split:
            pieces = _hipops.SplitRowsFn.apply(sizes, a, b, c)
            pa, pb = pieces[0:3], pieces[3:6]
        else:
            pa, pb = (a[0:4], a[4:4], a[4:10]), (b[0:4], b[4:4], b[4:10])
        # a: both ends used (one twice); b: only the last range; c: unused
        loss = (pa[0] ** 2).sum() + pa[0].sum() * 3 + (pa[2] * 0.5).sum() + (pb[2] ** 3).sum()
        loss.backward()
        return loss.detach(), a.grad, b.grad, c.grad

    got, ref = run(True), run(False)
    torch.testing.assert_close(got[0], ref[0])
    torch.testing.assert_close(got[1], ref[1])
    torch.testing.assert_close(got[2], ref[2])
    assert got[3] is None and ref[3] is None


def test_weighted_sum_matches_reference_formula():
    from trackertraincode.neuralnets import _hipops
    rng = np.random.default_rng(3)
    counts = [256, 256, 17, 1, 0, 300] * 6  # 36 terms: two forward launches
    vals0 = [torch.from_numpy(rng.normal(0, 1, n).astype(np.float32)).cuda() for n in counts]
    sws = [torch.from_numpy(rng.uniform(0.5, 2, n).astype(np.float32)).cuda() if i % 3 else None for i, n in enumerate(counts)]
    ws = [float(x) for x in rng.uniform(0.01, 2, len(counts))]
    vals = [v.clone().requires_grad_(True) for v in vals0]
    out = _hipops.WeightedSumFn.apply(ws, sws, 1.0 / 512, *vals)
    out.backward(torch.tensor(1.7, device="cuda"))
    ref_vals = [v.clone().double().requires_grad_(True) for v in vals0]
    ref = sum((v * (w * (s.double() if s is not None else 1.0))).sum() for v, w, s in zip(ref_vals, ws, sws)) / 512
    ref.backward(torch.tensor(1.7, device="cuda", dtype=torch.float64))
    assert abs(out.item() - ref.item()) <= 1e-6 * abs(ref.item()) + 1e-7
    for v, r in zip(vals, ref_vals):
        torch.testing.assert_close(v.grad, r.grad.float(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("cfg", ["full", "default", "rot6d"])
def test_batched_loss_launches_equal_single_launches(cfg, monkeypatch):
    """train.default_compute_loss issues all criterions' kernels as one ttk_loss_batch launch (and all their gradients as
    another, BatchedLossFn); with batching off every loss op is its own launch through its own autograd node.  Same kernels'
    bodies, same inputs: the loss, every per-sample value and the gradient reaching every prediction tensor must agree to
    the last bit or, where three gradients meet in one tensor, to the rounding of a different summation order."""
    import itertools

    import trackertraincode.train as train
    from trackertraincode.neuralnets import _hipops
    from util import build_net, load_golden, make_batches, script_args, train_script

    _, meta = load_golden(f"model_{cfg}.npz")
    S = train_script()
    torch.manual_seed(0)
    net = build_net(meta, "cuda").train()
    crit, _ = S.setup_losses(script_args(meta["flags"]), net)
    batches = make_batches(meta, "cuda")
    inputs = torch.concat([b["image"] for b in batches], dim=0)
    ids = torch.concat([b["coord_convention_id"] for b in batches], dim=0)
    with torch.no_grad():
        preds0 = net(inputs, ids)

    def run(batching):
        monkeypatch.setattr(_hipops, "_BATCHING", batching)
        preds = {k: (type(v)(v.value.detach().clone().requires_grad_(True)) if hasattr(v, "value") else v.detach().clone().requires_grad_(True))
                 for k, v in preds0.items()}
        leaves = {k: (v.value if hasattr(v, "value") else v) for k, v in preds.items()}
        launches = []
        orig = _hipops._hip.lib().call
        monkeypatch.setattr(_hipops._hip.lib(), "call", lambda name, *a: (launches.append(name), orig(name, *a))[1])
        loss, vals = train.default_compute_loss(preds, batches, 150, crit)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(_hipops._hip.lib(), "call", orig)
        by_name = train.concatenated_lossvals_by_name(itertools.chain.from_iterable(vals))
        return loss.detach(), {k: v[0] for k, v in by_name.items()}, {k: t.grad for k, t in leaves.items()}, launches

    la, va, ga, na = run(True)
    lb, vb, gb, nb = run(False)
    assert na.count("ttk_loss_batch") == 2 and not [n for n in na if n.startswith("ttk_loss_") and n != "ttk_loss_batch" and n in _hipops._hip.LOSS_BATCH_OPS]
    assert "ttk_loss_batch" not in nb and len([n for n in nb if n.startswith("ttk_loss_")]) >= 10
    assert torch.equal(la, lb)
    assert va.keys() == vb.keys() and all(torch.equal(va[k], vb[k]) for k in va)
    for k in ga:
        assert (ga[k] is None) == (gb[k] is None), k
        if ga[k] is not None:
            torch.testing.assert_close(ga[k], gb[k], rtol=1e-6, atol=1e-9, msg=k)


def test_criterion_that_post_processes_a_batched_loss_keeps_its_gradient(monkeypatch):
    """The reference's Criterion takes ANY callable (train.py:65-75).  One that does arithmetic on the per-sample values of a
    batched loss kernel (here: twice the rotation loss plus the size loss, as one term) would read memory the deferred launch
    has not filled and lose its gradient; default_compute_loss detects the unmatched record and evaluates the step's terms
    unbatched instead - same loss and gradients as with batching switched off, plus one RuntimeWarning."""
    import trackertraincode.train as train
    from trackertraincode.neuralnets import _hipops, losses
    from util import build_net, load_golden, make_batches

    _, meta = load_golden("model_default.npz")
    net = build_net(meta, "cuda").train()
    batches = make_batches(meta, "cuda")
    inputs = torch.concat([b["image"] for b in batches], dim=0)
    ids = torch.concat([b["coord_convention_id"] for b in batches], dim=0)
    with torch.no_grad():
        preds0 = net(inputs, ids)
    rot, size = losses.QuatPoseLoss("approx_distance"), losses.PoseSizeLoss("l2")
    crit = train.CriterionGroup([train.Criterion("combo", lambda p, b: 2.0 * rot(p, b) + size(p, b)[: b.meta.batchsize], 0.7),
                                 train.Criterion("xy", losses.PoseXYLoss("l2"), 1.0)])

    def run(batching):
        monkeypatch.setattr(_hipops, "_BATCHING", batching)
        monkeypatch.setattr(train, "_WARNED", set())
        preds = {k: (type(v)(v.value.detach().clone().requires_grad_(True)) if hasattr(v, "value") else v.detach().clone().requires_grad_(True))
                 for k, v in preds0.items()}
        leaves = {k: (v.value if hasattr(v, "value") else v) for k, v in preds.items()}
        loss, _ = train.default_compute_loss(preds, batches, 0, crit)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach(), {k: t.grad for k, t in leaves.items()}

    with pytest.warns(RuntimeWarning, match="post-processes"):
        la, ga = run(True)
    lb, gb = run(False)
    assert torch.equal(la, lb)
    assert ga["rot"] is not None and float(ga["rot"].abs().max()) > 0 and float(ga["coord"].abs().max()) > 0
    for k in ga:
        assert (ga[k] is None) == (gb[k] is None), k
        if ga[k] is not None:
            torch.testing.assert_close(ga[k], gb[k], rtol=1e-6, atol=1e-9, msg=k)
