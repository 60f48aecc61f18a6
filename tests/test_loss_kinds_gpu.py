"""GPU parity of the non-default loss kinds: the product's loss classes (HIP kernels ttk_loss_elem / ttk_loss_laplace /
ttk_loss_rot_geodesic through the C-ABI) against the reference's values and gradients in tests/golden/loss_kinds.npz."""
import numpy as np
import pytest
import torch

from test_loss_kinds import G, T

pytestmark = pytest.mark.gpu


def _cases():
    import trackertraincode.neuralnets.losses as L
    import trackertraincode.neuralnets.negloglikelihood as N

    c = {}
    for kind in ("l1", "smooth_l1"):
        c[f"xy/{kind}"] = (L.PoseXYLoss(kind), {"coord": "coord_p"}, {"coord": "coord_t"})
        c[f"size/{kind}"] = (L.PoseSizeLoss(kind), {"coord": "coord_p"}, {"coord": "coord_t"})
        c[f"box/{kind}"] = (L.BoxLoss(kind), {"roi": "roi_p"}, {"roi": "roi_t"})
        for dim in (2, 3):
            c[f"points{dim}/{kind}"] = (L.Points3dLoss(kind, pointdimension=dim, chin_weight=0.8, eye_weights=0.0), {"pt3d_68": "pts_p"}, {"pt3d_68": "pts_t"})
    c["rot/smooth_geodesic"] = (L.QuatPoseLoss("smooth_geodesic"), {"rot": "quat_p"}, {"pose": "quat_t"})
    for dist in ("laplace", "gaussian"):
        c[f"nllcoord_indep/{dist}"] = (N.CoordPoseNLLLoss(1.0, 0.5, dist), {"coord": "coord_p", "coord_scales": "coord_s"}, {"coord": "coord_t"})
    c["nllbox/laplace"] = (N.BoxNLLLoss(distribution="laplace"), {"roi": "roi_p", "roi_scales": "roi_s"}, {"roi": "roi_t"})
    for dim in (2, 3):
        c[f"nllpoints{dim}/laplace"] = (N.Points3dNLLLoss(0.8, 0.0, pointdimension=dim, distribution="laplace"), {"pt3d_68": "pts_p", "pt3d_68_scales": "pts_s"}, {"pt3d_68": "pts_t"})
    c["nllshape/laplace"] = (N.ShapeParamsNLLLoss("laplace"), {"shapeparam": "shape_p", "shapeparam_scales": "shape_s"}, {"shapeparam": "shape_t"})
    return c


def test_every_kind_matches_the_reference():
    from trackertraincode.neuralnets.rotrepr import QuatRepr

    cases = _cases()
    assert {k.rsplit("/values", 1)[0] for k in G.files if k.endswith("/values")} == set(cases)
    cot = T("cot").cuda()
    for name, (loss, pk, sk) in cases.items():
        if isinstance(loss, torch.nn.Module):
            loss = loss.cuda()
        leaves = {k: T(v).cuda().requires_grad_(True) for k, v in pk.items()}
        pred = {k: (QuatRepr(t) if k == "rot" else t) for k, t in leaves.items()}
        vals = loss(pred, {k: T(v).cuda() for k, v in sk.items()})
        assert vals.shape == cot.shape, name
        (vals * cot).sum().backward()
        geo = name.startswith("rot/")  # atan2 / sqrt chains in fp32: looser than the polynomial kinds
        np.testing.assert_allclose(vals.detach().cpu().numpy(), G[name + "/values"], rtol=2e-4 if geo else 2e-5, atol=1e-6, err_msg=name)
        for k, t in leaves.items():
            np.testing.assert_allclose(t.grad.cpu().numpy(), G[f"{name}/grad/{k}"], rtol=2e-3 if geo else 1e-4, atol=2e-5 if geo else 1e-6, err_msg=f"{name} {k}")


def test_unknown_kinds_raise():
    import trackertraincode.neuralnets.losses as L
    import trackertraincode.neuralnets.negloglikelihood as N

    with pytest.raises(KeyError):
        L.PoseXYLoss("huber")
    with pytest.raises(KeyError):
        L.QuatPoseLoss("chordal")
    with pytest.raises(KeyError):
        N.BoxNLLLoss(distribution="cauchy")


def test_kinds_inside_a_loss_batch():
    """The non-default kinds launch one by one next to the deferred default ops of train.default_compute_loss's batch."""
    import trackertraincode.neuralnets.losses as L
    from trackertraincode.neuralnets import _hipops

    p, t = T("roi_p").cuda().requires_grad_(True), T("roi_t").cuda()
    with _hipops.loss_batch() as b:
        v_l1 = L.BoxLoss("l1")({"roi": p}, {"roi": t})
        v_l2 = L.BoxLoss("l2")({"roi": p.detach()}, {"roi": t})
        b.flush()
    np.testing.assert_allclose(v_l1.detach().cpu().numpy(), G["box/l1/values"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(v_l2.cpu().numpy(), ((G["in/roi_p"] - G["in/roi_t"]) ** 2).mean(-1), rtol=2e-5, atol=1e-7)
