"""Uncertainty heads and negative-log-likelihood losses with the reference's names (reference:
neuralnets/negloglikelihood.py).  Parameter containers keep the reference's state-dict names; the
loss values and their gradients come from HIP kernels (csrc/losses.hip, csrc/loss_math.h) through
_hipops - CUDA tensors only, no PyTorch fallback.
"""
from __future__ import annotations

from typing import Literal

import torch
import torch.nn as nn
from torch import Tensor

from . import _hipops
from .math import inv_smoothclip0, smoothclip0

make_positive = smoothclip0
inv_make_positive = inv_smoothclip0

SimpleDistributionSwitch = Literal["gaussian", "laplace"]


def _dist_fn(distribution):
    """DISTRIBUTION_CLASS_MAP of the reference (:68-69): Normal or Laplace, both parameterised (location, scale)."""
    return {"gaussian": _hipops.NormalNllFn, "laplace": _hipops.LaplaceNllFn}[distribution]


class Neck(nn.Module):
    """Linear layer with one extra output that acts as a common positive multiplier (reference :22-35)."""

    def __init__(self, num_in_features, num_out_features):
        super().__init__()
        self.num_in_features, self.num_out_features = num_in_features, num_out_features
        self.lin = nn.Linear(num_in_features, num_out_features + 1)
        self.lin.bias.data[...] = inv_make_positive(torch.ones((num_out_features + 1)))

    def set_biases(self, x: Tensor):
        self.lin.bias.data[..., 1:] = x

    def forward(self, x: Tensor):
        y = self.lin(x)
        return y[..., 1:], make_positive(y[..., :1])


class FeaturesAsDiagonalScale(nn.Module):
    def __init__(self, num_in_features, num_out_features):
        super().__init__()
        self.neck = Neck(num_in_features, num_out_features)
        self.eps = torch.tensor(1.0e-6)

    def forward(self, x: Tensor):
        y, mult = self.neck(x)
        return make_positive(y) * mult + self.eps


class DiagonalScaleParameter(nn.Module):
    """Trainable, input-independent positive scales, initialised to 1 (reference :50-65)."""

    def __init__(self, num_out_features):
        super().__init__()
        self.hidden_scale = nn.Parameter(inv_make_positive(torch.ones((num_out_features + 1,))))
        self.eps = torch.tensor(1.0e-6)

    def forward(self):
        if self.hidden_scale.is_cuda:
            return _hipops.DiagScaleFn.apply(self.hidden_scale)
        return make_positive(self.hidden_scale[:1]) * make_positive(self.hidden_scale[1:]) + self.eps


def _fill_triangular_matrix(dim: int, z: Tensor) -> Tensor:
    """z = [diagonal..., strictly-lower entries row by row] -> lower-triangular (dim x dim) (reference :187-211)."""
    assert dim == 3, "only the 3x3 case is used by the pose network"
    zero = torch.zeros_like(z[..., 0])
    rows = (z[..., 0], zero, zero, z[..., 3], z[..., 1], zero, z[..., 4], z[..., 5], z[..., 2])
    return torch.stack(rows, dim=-1).view(*z.shape[:-1], 3, 3)


class FeaturesAsTriangularScale(nn.Module):
    """Features -> Cholesky-like factor of a 3x3 covariance (reference :214-242).  In the training step
    the arithmetic runs in the fused heads kernel (head_math.h: tri_scale_fwd/bwd)."""

    def __init__(self, num_in_features, dim):
        super().__init__()
        self.dim = dim
        self.num_matrix_params = (dim * (dim + 1)) // 2
        self.neck = Neck(num_in_features, self.num_matrix_params)
        bias_init = inv_make_positive(torch.ones((self.num_matrix_params)))
        bias_init[self.dim:] = 0.0
        self.neck.set_biases(bias_init)
        min_diag = torch.full((self.num_matrix_params,), 1.0e-6)
        min_diag[self.dim:] = 0.0
        self.register_buffer("min_diag", min_diag)

    def forward(self, x: Tensor):
        y, mult = self.neck(x)
        z = torch.cat((make_positive(y[..., : self.dim]), y[..., self.dim:]), dim=-1)
        return _fill_triangular_matrix(self.dim, mult * z + self.min_diag)


# ---------------------------------------------------------------------------------------------
# losses (callable(pred, sample) -> per-sample tensor [n])
# ---------------------------------------------------------------------------------------------
def _as_quat(rot):
    return rot.as_quat() if hasattr(rot, "as_quat") else rot


class CoordPoseNLLLoss(nn.Module):
    """Independent Normal per coordinate, weighted [xy/2, xy/2, size] and averaged over the three (reference :72-97;
    the training script uses CorrelatedCoordPoseNLLLoss instead).  `coord_scales` are per-coordinate standard
    deviations [n, 3]."""

    def __init__(self, xy_weight: float, head_size_weight: float, distribution: SimpleDistributionSwitch = "gaussian"):
        super().__init__()
        self._fn = _dist_fn(distribution)
        self._w = (xy_weight / 2.0, xy_weight / 2.0, float(head_size_weight))  # host copy: no device read per step
        self.register_buffer("weights", torch.as_tensor(self._w, dtype=torch.float32))

    def __call__(self, preds, sample):
        mu, sigma, x = preds["coord"], preds["coord_scales"], sample["coord"]
        if sigma.shape != mu.shape:
            raise ValueError(f"CoordPoseNLLLoss: coord_scales {tuple(sigma.shape)} must be per-coordinate like coord {tuple(mu.shape)}")
        # -mean_d w_d log N(x_d; mu_d, sigma_d): one single-column launch of the Normal kernel per coordinate
        terms = [self._fn.apply(mu[:, d:d + 1], sigma[:, d:d + 1], x[:, d:d + 1], False, 0, 1.0, 1.0) for d in range(3)]
        return (terms[0] * self._w[0] + terms[1] * self._w[1] + terms[2] * self._w[2]) / 3.0


class MixWithUniformProbability(nn.Module):
    """log(0.999 p + 0.001 / volume) (reference :100-110); folded into the nllrot / nllcoord kernels."""

    def __init__(self, state_space_volume):
        super().__init__()
        self.register_buffer("log_uniform_prob", -torch.as_tensor([state_space_volume]).log())
        self.register_buffer("log_weights", torch.as_tensor([[0.999, 0.001]]).log())

    def __call__(self, log_prob):
        both = torch.stack((log_prob, torch.broadcast_to(self.log_uniform_prob, log_prob.shape)), dim=-1)
        return torch.logsumexp(both + self.log_weights, dim=-1)


class CorrelatedCoordPoseNLLLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.uniform_mixing = MixWithUniformProbability(4.0)  # [-1,1] x [-1,1] x [0,1]

    def __call__(self, preds, sample):
        return _hipops.apply(_hipops.NllCoordFn, preds["coord"], sample["coord"], preds["coord_scales"])


class BoxNLLLoss(nn.Module):
    def __init__(self, dataname="roi", distribution: SimpleDistributionSwitch = "gaussian"):
        super().__init__()
        self._fn = _dist_fn(distribution)
        self.dataname = dataname

    def __call__(self, pred, sample):
        return _hipops.apply(self._fn, pred[self.dataname], pred[self.dataname + "_scales"], sample[self.dataname], False, 0, 1.0, 1.0)


class Points3dNLLLoss(nn.Module):
    def __init__(self, chin_weight, eye_weight, pointdimension: int = 3, distribution: SimpleDistributionSwitch = "gaussian"):
        super().__init__()
        self._fn = _dist_fn(distribution)
        from .losses import point_weights

        self.register_buffer("pointweights", point_weights(chin_weight, eye_weight))
        self.chin_weight, self.eye_weight, self.pointdimension = float(chin_weight), float(eye_weight), pointdimension

    def __call__(self, preds, sample):
        return _hipops.apply(self._fn, preds["pt3d_68"], preds["pt3d_68_scales"], sample["pt3d_68"], True,
                                         self.pointdimension, self.chin_weight, self.eye_weight)


class ShapeParamsNLLLoss(nn.Module):
    def __init__(self, distribution: SimpleDistributionSwitch = "gaussian"):
        super().__init__()
        self._fn = _dist_fn(distribution)

    def __call__(self, preds, sample):
        return _hipops.apply(self._fn, preds["shapeparam"], preds["shapeparam_scales"], sample["shapeparam"], False, 0, 1.0, 1.0)


class QuatPoseNLLLoss(nn.Module):
    """Gaussian in the tangent space of the predicted rotation, mixed with a uniform floor over the ball
    of radius pi (reference :245-274)."""

    def __init__(self):
        super().__init__()
        self.uniform_mixing = MixWithUniformProbability(torch.pi ** 4 * 4.0 / 3.0)

    def __call__(self, preds, sample):
        return _hipops.apply(_hipops.NllRotFn, _as_quat(preds["rot"]), sample["pose"], preds["pose_scales_tril"])
