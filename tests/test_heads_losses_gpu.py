"""Per-op GPU parity: fused heads kernel and every loss kernel (through the C-ABI) against the CPU
oracle evaluated in float64.  Tolerances are fp32-rounding level."""
import os

import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.synth import make_labels, make_state
from util import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _head_state(unc, pt, dtype):
    shapes = {k: v for k, v in R.state_shapes(pt, unc).items() if not k.startswith("convnet.")}
    st = {}
    for k, v in make_state(shapes, 3).items():
        t = torch.from_numpy(np.array(v))
        t = t.to(dtype) if t.is_floating_point() else t
        st[k] = t.requires_grad_(True) if not R.is_buffer(k) else t
    return st


@pytest.mark.parametrize("unc,pt", [(True, True), (False, True), (False, False)])
def test_heads_fwd_bwd(unc, pt):
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    rng = np.random.default_rng(0)
    B = 37
    feat = np.abs(rng.standard_normal((B, 1024))).astype(np.float32) * 0.5
    ids = rng.integers(0, 8, B).astype(np.int32)
    st = _head_state(unc, pt, torch.float64)
    f64 = torch.from_numpy(feat).double().requires_grad_(True)
    out = R.heads_forward(st, f64, torch.from_numpy(ids), enable_point_head=pt, enable_uncertainty=unc, training=True)
    ups = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape))) for k, v in out.items()}
    sum((out[k] * ups[k]).sum() for k in out).backward()

    net = NetworkWithPointHead(enable_point_head=pt, enable_uncertainty=unc).to(DEV)
    sd = net.state_dict()
    for k, v in st.items():
        sd[k].copy_(v.detach().float())
    fg = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    mine = net._heads_hip(fg, torch.from_numpy(ids).to(DEV))
    loss = 0
    for k in out:
        v = mine[k].value if hasattr(mine[k], "value") else mine[k]
        assert rel(v.detach(), out[k].detach()) < 2e-6, k
        loss = loss + (v * ups[k].float().to(DEV)).sum()
    loss.backward()
    assert rel(fg.grad, f64.grad) < 5e-6
    for k, p in net.named_parameters():
        if k.startswith("convnet.") or st[k].grad is None:
            continue
        assert rel(p.grad, st[k].grad) < 5e-6, k


def _loss_inputs(n, seed=0):
    rng = np.random.default_rng(seed)
    lab = make_labels(n, seed=5)
    unit = lambda a: a / np.linalg.norm(a, axis=-1, keepdims=True)
    L = np.zeros((n, 3, 3))
    L[:, [0, 1, 2], [0, 1, 2]] = rng.uniform(0.3, 1.5, (n, 3))
    L[:, 1, 0], L[:, 2, 0], L[:, 2, 1] = rng.normal(0, 0.3, (3, n))
    pred = {
        "rot": unit(rng.standard_normal((n, 4))), "unnormalized_quat": rng.standard_normal((n, 4)),
        "coord": lab["coord"] + 0.2 * rng.standard_normal((n, 3)), "roi": lab["roi"] + 0.2 * rng.standard_normal((n, 4)),
        "pt3d_68": lab["pt3d_68"] + 0.3 * rng.standard_normal((n, 68, 3)), "shapeparam": 0.5 * rng.standard_normal((n, 50)),
        "pose_scales_tril": L, "coord_scales": L[::-1].copy(), "roi_scales": rng.uniform(0.3, 2, (n, 4)),
        "pt3d_68_scales": rng.uniform(0.3, 2, (n, 68, 3)), "shapeparam_scales": rng.uniform(0.3, 2, (n, 50)),
    }
    return pred, lab


def test_every_loss_kernel():
    from util import script_args, train_script
    import trackertraincode.neuralnets.losses as LS
    import trackertraincode.neuralnets.negloglikelihood as NLL

    n = 53
    pred0, lab = _loss_inputs(n)
    gmm = R.ShapeGmm(os.path.join(GOLDEN, "shapeparams_gmm.npz"))
    cases = [
        ("rot", LS.QuatPoseLoss("approx_distance"), R.loss_rot), ("xy", LS.PoseXYLoss("l2"), R.loss_xy),
        ("sz", LS.PoseSizeLoss("l2"), R.loss_sz), ("box", LS.BoxLoss("l2"), R.loss_box),
        ("points3d", LS.Points3dLoss("l2", chin_weight=0.8, eye_weights=0.0), R.loss_points3d),
        ("points2d", LS.Points3dLoss("l2", pointdimension=2, chin_weight=0.8, eye_weights=0.0), lambda p, s: R.loss_points3d(p, s, 2)),
        ("shp_l2", LS.ShapeParameterLoss(), R.loss_shp_l2), ("quatreg", LS.QuaternionNormalizationSoftConstraint(), R.loss_quatreg),
        ("gmm", LS.ShapePlausibilityLoss(), gmm), ("nllrot", NLL.QuatPoseNLLLoss(), R.loss_nllrot),
        ("nllcoord", NLL.CorrelatedCoordPoseNLLLoss(), R.loss_nllcoord), ("nllbox", NLL.BoxNLLLoss(), R.loss_nllbox),
        ("nllpoints3d", NLL.Points3dNLLLoss(0.8, 0.0), R.loss_nllpoints3d),
        ("nllpoints2d", NLL.Points3dNLLLoss(0.8, 0.0, pointdimension=2), lambda p, s: R.loss_nllpoints3d(p, s, 2)),
        # reference negloglikelihood.py:169-177 and :72-97 restated with torch.distributions (fp64)
        ("nllshape", NLL.ShapeParamsNLLLoss(), lambda p, s: -torch.distributions.Normal(p["shapeparam"], p["shapeparam_scales"]).log_prob(s["shapeparam"]).mean(-1)),
        ("nllcoordpose", NLL.CoordPoseNLLLoss(0.6, 0.25),
         lambda p, s: -torch.distributions.Normal(p["coord"], p["coord_scales"]).log_prob(s["coord"]).mul(torch.tensor([0.3, 0.3, 0.25], dtype=torch.float64)[None, :]).mean(-1)),
    ]
    gv = np.random.default_rng(9).standard_normal(n)
    diag_scales = np.random.default_rng(11).uniform(0.3, 2, (n, 3))  # CoordPoseNLLLoss: independent per-coordinate scales
    for name, mine, ref in cases:
        pred = dict(pred0, coord_scales=diag_scales) if name == "nllcoordpose" else pred0
        p64 = {k: torch.from_numpy(np.ascontiguousarray(v)).double().requires_grad_(True) for k, v in pred.items()}
        s64 = {k: torch.from_numpy(v).double() for k, v in lab.items()}
        v_ref = ref(p64, s64)
        (v_ref.double() * torch.from_numpy(gv)).sum().backward()
        pg = {k: torch.from_numpy(np.ascontiguousarray(v)).float().to(DEV).requires_grad_(True) for k, v in pred.items()}
        sg = {k: torch.from_numpy(v).float().to(DEV) for k, v in lab.items()}
        v = mine(pg, sg)
        assert v.shape == (n,)
        (v * torch.from_numpy(gv).float().to(DEV)).sum().backward()
        assert rel(v.detach(), v_ref.detach()) < 3e-6, name
        for k in pred:
            if p64[k].grad is None:
                assert pg[k].grad is None or float(pg[k].grad.abs().max()) == 0.0, (name, k)
            else:
                assert rel(pg[k].grad, p64[k].grad) < 1e-5, (name, k, rel(pg[k].grad, p64[k].grad))
