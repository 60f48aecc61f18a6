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

SimpleLossSwitch = Literal["l2", "l1", "smooth_l1"]
SimpleRotLossSwitch = Literal["approx_distance", "smooth_geodesic"]


def _check_kind(kind):
    """The keys of the reference's LOSS_OBJECT_MAP (:16-21): MSELoss, L1Loss, SmoothL1Loss(beta=0.01), all reduction="none"."""
    if kind not in _hipops.ELEM_KINDS:
        raise KeyError(kind)
    return kind


class _ColumnWeights:
    """Device copies of the per-column weight vector of a non-"l2" loss (ttk_loss_elem: v = sum_d colw[d] f(p_d - t_d)), one per device."""

    def __init__(self, values):
        self._host = torch.as_tensor(values, dtype=torch.float32).contiguous()
        self._dev = {}

    def on(self, device):
        key = (device.type, device.index)
        if key not in self._dev:
            self._dev[key] = self._host.to(device)
        return self._dev[key]


def _window_weights(D, c0, nc):
    w = np.zeros((D,), dtype=np.float32)
    w[c0:c0 + nc] = 1.0 / nc  # .mean(dim=-1) over the window
    return w


def point_weights(chin_weight=1.0, eye_weight=0.0) -> torch.Tensor:
    """Per-landmark weights: jaw line except the chin tip, and the eye-lid points (reference :139-142)."""
    w = np.ones((68,), dtype=np.float32)
    w[kpts68.chin_left[:-1]] = chin_weight
    w[kpts68.chin_right[1:]] = chin_weight
    w[kpts68.eye_not_corners] = eye_weight
    return torch.from_numpy(w)


class QuatPoseLoss:
    def __init__(self, loss: SimpleRotLossSwitch, prefix=""):
        # reference :35-39: "approx_distance" = torchquaternion.distance, "smooth_geodesic" = smooth_geodesic_distance (:24-32)
        self._fn = {"approx_distance": _hipops.RotLossFn, "smooth_geodesic": _hipops.RotGeodesicFn}[loss]
        self._prefix = prefix

    def __call__(self, pred, sample):
        quat = pred[self._prefix + "rot"]
        return _hipops.apply(self._fn, quat.value if hasattr(quat, "value") else quat, sample["pose"])


class _CoordWindowLoss:
    """loss_obj(coord[..., window], target[..., window]) averaged over the window (reference :67-88)."""

    _c0, _nc = 0, 3

    def __init__(self, loss: SimpleLossSwitch, prefix=""):
        self._kind = _check_kind(loss)
        self._prefix = prefix
        self._colw = _ColumnWeights(_window_weights(3, self._c0, self._nc))

    def __call__(self, pred, sample):
        p, t = pred[self._prefix + "coord"], sample["coord"]
        if self._kind == "l2":
            return _hipops.mse_cols(p, t, self._c0, self._nc)
        return _hipops.apply(_hipops.ElemLossFn, p, t, self._colw.on(p.device), _hipops.ELEM_KINDS[self._kind])


class PoseSizeLoss(_CoordWindowLoss):
    _c0, _nc = 2, 1


class PoseXYLoss(_CoordWindowLoss):
    _c0, _nc = 0, 2


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
        self._kind = _check_kind(loss)
        assert pointdimension in (2, 3)
        self._prefix, self.pointdimension = prefix, pointdimension
        self.chin_weight, self.eye_weights = float(chin_weight), float(eye_weights)
        self.register_buffer("pointweights", point_weights(chin_weight, eye_weights))
        # mean_p w_p sum_{d < dim} f(.) as one weight per (point, coordinate)
        w = (self.pointweights[:, None] / 68.0) * torch.tensor([1.0, 1.0, 1.0 if pointdimension == 3 else 0.0])
        self._colw = _ColumnWeights(w.reshape(-1))

    def _eval_on_points(self, pred, target):
        assert target.shape == pred.shape, f"Mismatch {target.shape} vs {pred.shape}"
        assert target.shape[1] == 68 and target.shape[2] == 3
        if self._kind != "l2":
            return _hipops.apply(_hipops.ElemLossFn, pred, target, self._colw.on(pred.device), _hipops.ELEM_KINDS[self._kind])
        return _hipops.apply(_hipops.PointsLossFn, pred, target, self.pointdimension, self.chin_weight, self.eye_weights)

    def forward(self, pred, sample):
        return self._eval_on_points(pred[self._prefix + "pt3d_68"], sample["pt3d_68"])


class BoxLoss:
    def __init__(self, loss: SimpleLossSwitch, dataname="roi"):
        self._kind = _check_kind(loss)
        self.dataname = dataname
        self._colw = _ColumnWeights(_window_weights(4, 0, 4))

    def __call__(self, pred, sample):
        p, t = pred[self.dataname], sample[self.dataname]
        if self._kind != "l2":
            return _hipops.apply(_hipops.ElemLossFn, p, t, self._colw.on(p.device), _hipops.ELEM_KINDS[self._kind])
        return _hipops.apply(_hipops.MseRowsFn, p, t)


class Rot6dReprLoss:
    """0.75 - 0.25 tr(R tomatrix(target)^T) (reference :53-58, torch6drotation.py:68-72)."""

    def __call__(self, pred_batch, target_batch):
        pred = pred_batch["rot"]
        return _hipops.apply(_hipops.Rot6dLossFn, pred.value if hasattr(pred, "value") else pred, target_batch["pose"])


class Rot6dNormalizationSoftConstraint:
    """mean((M M^T - I_2)^2) of the raw 6D features (reference :61-64, torch6drotation.py:20-24)."""

    def __call__(self, pred_batch, target_batch):
        return _hipops.apply(_hipops.Ortho6dFn, pred_batch["unnormalized_6drepr"])
