#!/usr/bin/env python3
"""Golden vectors of the NON-DEFAULT kinds of the reference's loss switches, produced by the reference's own loss classes imported through
ref_shims (build container only; /root/reference does not travel):

  neuralnets/losses.py:16-21   LOSS_OBJECT_MAP "l1" / "smooth_l1" (beta 0.01) in PoseXYLoss :79-88, PoseSizeLoss :67-76, BoxLoss :163-173,
                               Points3dLoss :128-160 (dimension 2 and 3, chin / eye weights of the training script)
  neuralnets/losses.py:24-39   smooth_geodesic_distance in QuatPoseLoss :42-50
  neuralnets/negloglikelihood.py:68-69  distribution="laplace" in CoordPoseNLLLoss :72-97, BoxNLLLoss :129-142, Points3dNLLLoss :145-166,
                               ShapeParamsNLLLoss :169-177

Stored per case: the per-sample values and the gradient of sum(values * cot) with respect to every prediction tensor (cot = a fixed random
cotangent).  Inputs are stored too.  -> tests/golden/loss_kinds.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shims  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
torch = ref_shims.install(gmm_npz=os.path.join(GOLD, "shapeparams_gmm.npz"))
import trackertraincode.neuralnets.losses as RL  # noqa: E402
import trackertraincode.neuralnets.negloglikelihood as RN  # noqa: E402
from trackertraincode.neuralnets.rotrepr import QuatRepr  # noqa: E402

rng = np.random.default_rng(77)
n = 24
f32 = np.float32


def near(shape, scale, tiny_frac=0.3):
    """Differences on both sides of smooth_l1's beta = 0.01 (and a few exact zeros for l1's sign(0))."""
    d = rng.standard_normal(shape) * scale
    tiny = rng.random(shape) < tiny_frac
    d = np.where(tiny, rng.standard_normal(shape) * 0.004, d)
    d.reshape(-1)[:2] = 0.0
    return d.astype(f32)


inp = {}
inp["coord_t"] = np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(0.2, 0.8, (n, 1))], -1).astype(f32)
inp["coord_p"] = inp["coord_t"] + near((n, 3), 0.05)
inp["roi_t"] = rng.uniform(-0.8, 0.8, (n, 4)).astype(f32)
inp["roi_p"] = inp["roi_t"] + near((n, 4), 0.05)
inp["pts_t"] = (rng.standard_normal((n, 68, 3)) * 0.3).astype(f32)
inp["pts_p"] = inp["pts_t"] + near((n, 68, 3), 0.03)
inp["shape_t"] = rng.standard_normal((n, 50)).astype(f32)
inp["shape_p"] = inp["shape_t"] + near((n, 50), 0.3)
qt = rng.standard_normal((n, 4))
qt /= np.linalg.norm(qt, axis=-1, keepdims=True)
ang = np.concatenate([rng.uniform(0.0, 0.015, n // 3), rng.uniform(0.02, 1.5, n - n // 3)])  # below and above the 1-degree zone
ax = rng.standard_normal((n, 3))
ax /= np.linalg.norm(ax, axis=-1, keepdims=True)
dq = np.concatenate([ax * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], -1)
import trackertraincode.neuralnets.torchquaternion as TQ  # noqa: E402

qp = TQ.mult(torch.from_numpy(qt), torch.from_numpy(dq)).numpy()
qp[::2] *= -1.0  # both signs of the same rotation
inp["quat_t"], inp["quat_p"] = qt.astype(f32), qp.astype(f32)
inp["coord_s"] = rng.uniform(0.02, 0.3, (n, 3)).astype(f32)
inp["roi_s"] = rng.uniform(0.02, 0.3, (n, 4)).astype(f32)
inp["pts_s"] = rng.uniform(0.01, 0.2, (n, 68, 3)).astype(f32)
inp["shape_s"] = rng.uniform(0.1, 1.5, (n, 50)).astype(f32)
inp["cot"] = rng.uniform(0.5, 1.5, n).astype(f32)
out = {"in/" + k: v for k, v in inp.items()}


def run(name, loss, pred_keys, pred, sample):
    leaves = {}
    for k in pred_keys:
        v = pred[k]
        t = (v.value if hasattr(v, "value") else v).clone().requires_grad_(True)
        leaves[k] = t
        pred[k] = QuatRepr(t) if hasattr(v, "value") else t
    vals = loss(pred, sample)
    assert vals.shape == (n,), (name, vals.shape)
    (vals * torch.from_numpy(inp["cot"])).sum().backward()
    out[f"{name}/values"] = vals.detach().numpy()
    for k, t in leaves.items():
        out[f"{name}/grad/{k}"] = t.grad.numpy()


T = lambda k: torch.from_numpy(inp[k].copy())
for kind in ("l1", "smooth_l1"):
    run(f"xy/{kind}", RL.PoseXYLoss(kind), ["coord"], {"coord": T("coord_p")}, {"coord": T("coord_t")})
    run(f"size/{kind}", RL.PoseSizeLoss(kind), ["coord"], {"coord": T("coord_p")}, {"coord": T("coord_t")})
    run(f"box/{kind}", RL.BoxLoss(kind), ["roi"], {"roi": T("roi_p")}, {"roi": T("roi_t")})
    for dim in (2, 3):
        run(f"points{dim}/{kind}", RL.Points3dLoss(kind, pointdimension=dim, chin_weight=0.8, eye_weights=0.0), ["pt3d_68"],
            {"pt3d_68": T("pts_p")}, {"pt3d_68": T("pts_t")})
run("rot/smooth_geodesic", RL.QuatPoseLoss("smooth_geodesic"), ["rot"], {"rot": QuatRepr(T("quat_p"))}, {"pose": T("quat_t")})
run("nllcoord_indep/laplace", RN.CoordPoseNLLLoss(1.0, 0.5, "laplace"), ["coord", "coord_scales"],
    {"coord": T("coord_p"), "coord_scales": T("coord_s")}, {"coord": T("coord_t")})
run("nllcoord_indep/gaussian", RN.CoordPoseNLLLoss(1.0, 0.5, "gaussian"), ["coord", "coord_scales"],
    {"coord": T("coord_p"), "coord_scales": T("coord_s")}, {"coord": T("coord_t")})
run("nllbox/laplace", RN.BoxNLLLoss(distribution="laplace"), ["roi", "roi_scales"],
    {"roi": T("roi_p"), "roi_scales": T("roi_s")}, {"roi": T("roi_t")})
for dim in (2, 3):
    run(f"nllpoints{dim}/laplace", RN.Points3dNLLLoss(0.8, 0.0, pointdimension=dim, distribution="laplace"), ["pt3d_68", "pt3d_68_scales"],
        {"pt3d_68": T("pts_p"), "pt3d_68_scales": T("pts_s")}, {"pt3d_68": T("pts_t")})
run("nllshape/laplace", RN.ShapeParamsNLLLoss("laplace"), ["shapeparam", "shapeparam_scales"],
    {"shapeparam": T("shape_p"), "shapeparam_scales": T("shape_s")}, {"shapeparam": T("shape_t")})

np.savez_compressed(os.path.join(GOLD, "loss_kinds.npz"), **out)
print("loss_kinds.npz:", len(out), "entries")
