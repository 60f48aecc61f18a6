"""Loss callables with the reference's names (reference: neuralnets/losses.py): `loss(pred, sample)`
returns one value per sample.  Values and gradients come from HIP kernels (csrc/losses.hip)."""
from __future__ import annotations

import os
from typing import Literal

import numpy as np
import torch
import torch.nn as nn

from . import _hipops
from ..facemodel import keypoints68 as kpts68
from .modelcomponents import FACEMODEL_DIR, GaussianMixture

SimpleLossSwitch = Literal["l2", "smooth_l1"]
SimpleRotLossSwitch = Literal["approx_distance", "smooth_geodesic"]


def _l2_only(kind):
    if kind != "l2":
        raise NotImplementedError(f"loss={kind!r}: only 'l2' is built (the training script uses nothing else)")


def point_weights(chin_weight=1.0, eye_weight=0.0) -> torch.Tensor:
    """Per-landmark weights: jaw line except the chin tip, and the eye-lid points (reference :139-142)."""
    w = np.ones((68,), dtype=np.float32)
    w[kpts68.chin_left[:-1]] = chin_weight
    w[kpts68.chin_right[1:]] = chin_weight
    w[kpts68.eye_not_corners] = eye_weight
    return torch.from_numpy(w)


class QuatPoseLoss:
    def __init__(self, loss: SimpleRotLossSwitch, prefix=""):
        if loss != "approx_distance":
            raise NotImplementedError("only the 'approx_distance' rotation loss is built (scripts/train_poseestimator.py:172)")
        self._prefix = prefix

    def __call__(self, pred, sample):
        quat = pred[self._prefix + "rot"]
        return _hipops.apply(_hipops.RotLossFn, quat.value if hasattr(quat, "value") else quat, sample["pose"])


class PoseSizeLoss:
    def __init__(self, loss: SimpleLossSwitch, prefix=""):
        _l2_only(loss)
        self._prefix = prefix

    def __call__(self, pred, sample):
        return _hipops.mse_cols(pred[self._prefix + "coord"], sample["coord"], 2, 1)


class PoseXYLoss:
    def __init__(self, loss: SimpleLossSwitch, prefix=""):
        _l2_only(loss)
        self._prefix = prefix

    def __call__(self, pred, sample):
        return _hipops.mse_cols(pred[self._prefix + "coord"], sample["coord"], 0, 2)


class ShapeParameterLoss:
    def eval_on_params(self, pred, target):
        return _hipops.apply(_hipops.MseRowsFn, pred, target)

    def __call__(self, pred, sample):
        return self.eval_on_params(pred["shapeparam"], sample["shapeparam"])


class ShapePlausibilityLoss(nn.Module):
    """-log p_GMM(shapeparam) * (0.001 / n_components), evaluated in float64 (reference :100-113)."""

    def __init__(self):
        super().__init__()
        self.gmm = GaussianMixture.from_npz(os.path.join(FACEMODEL_DIR, "shapeparams_gmm.npz"))
        self.register_buffer("fudge_factor", torch.as_tensor(0.001 / self.gmm.n_components))
        g = self.gmm
        ck = torch.log(g.weights) + torch.log(g.scales_inv).sum(-1) - g.norm_constant
        self.register_buffer("_ck", ck.to(torch.float64).contiguous(), persistent=False)
        self.register_buffer("_mu", g.means.to(torch.float64).contiguous(), persistent=False)
        self.register_buffer("_sinv", g.scales_inv.to(torch.float64).contiguous(), persistent=False)

    def forward(self, pred, sample):
        x = pred["shapeparam"]
        if self._ck.device != x.device:
            self.to(x.device)
        return _hipops.apply(_hipops.GmmNllFn, x, self._ck, self._mu, self._sinv, 0.001 / self.gmm.n_components)


class QuaternionNormalizationSoftConstraint:
    def __init__(self, prefix=""):
        self._prefix = prefix

    def __call__(self, pred, sample):
        q = pred[self._prefix + "unnormalized_quat"]
        assert q.dim() == 2 and q.shape[-1] == 4
        return _hipops.apply(_hipops.QuatRegFn, q)


class Points3dLoss(nn.Module):
    def __init__(self, loss: SimpleLossSwitch, pointdimension: int = 3, chin_weight=1.0, eye_weights=0.0, prefix=""):
        super().__init__()
        _l2_only(loss)
        assert pointdimension in (2, 3)
        self._prefix, self.pointdimension = prefix, pointdimension
        self.chin_weight, self.eye_weights = float(chin_weight), float(eye_weights)
        self.register_buffer("pointweights", point_weights(chin_weight, eye_weights))

    def _eval_on_points(self, pred, target):
        assert target.shape == pred.shape, f"Mismatch {target.shape} vs {pred.shape}"
        assert target.shape[1] == 68 and target.shape[2] == 3
        return _hipops.apply(_hipops.PointsLossFn, pred, target, self.pointdimension, self.chin_weight, self.eye_weights)

    def forward(self, pred, sample):
        return self._eval_on_points(pred[self._prefix + "pt3d_68"], sample["pt3d_68"])


class BoxLoss:
    def __init__(self, loss: SimpleLossSwitch, dataname="roi"):
        _l2_only(loss)
        self.dataname = dataname

    def __call__(self, pred, sample):
        return _hipops.apply(_hipops.MseRowsFn, pred[self.dataname], sample[self.dataname])


class Rot6dReprLoss:
    """0.75 - 0.25 tr(R tomatrix(target)^T) (reference :53-58, torch6drotation.py:68-72)."""

    def __call__(self, pred_batch, target_batch):
        pred = pred_batch["rot"]
        return _hipops.apply(_hipops.Rot6dLossFn, pred.value if hasattr(pred, "value") else pred, target_batch["pose"])


class Rot6dNormalizationSoftConstraint:
    """mean((M M^T - I_2)^2) of the raw 6D features (reference :61-64, torch6drotation.py:20-24)."""

    def __call__(self, pred_batch, target_batch):
        return _hipops.apply(_hipops.Ortho6dFn, pred_batch["unnormalized_6drepr"])
