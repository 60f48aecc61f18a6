"""Building blocks of the pose network with the reference's class names (reference:
neuralnets/modelcomponents.py).  On the MI355X the arithmetic of DeformableHeadKeypoints,
rigid_transformation_25d and LocalToGlobalCoordinateOffset runs inside the fused heads kernel
(csrc/heads.hip); the classes here own the parameters/buffers under the reference's state-dict names
and provide the plain-torch form used for CPU eval/export and by evaluation scripts.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Type

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from .math import smoothclip0
from .rotrepr import RotationRepr

FACEMODEL_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "facemodel")


def set_bn_momentum(model: nn.Module, momentum):
    for m in model.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = momentum


class BlurPool2D(nn.Module):
    """Blur + downsample (reference :187-205: kornia's blur pooling with an explicit channel count): a depthwise
    convolution with the normalised binomial kernel (kernel_size 3: outer([1,2,1],[1,2,1]) / 16), zero padding
    (kernel_size - 1) // 2 and the given stride.  Owns the reference's buffer `kernel` [k, k]; `depthwise_weight()` is the
    same kernel as a [C,1,k,k] depthwise weight - the form csrc/dwconv_tiled.hip's kernels take on the MI355X (the strided
    MobileNet blocks run the blur as one more 3x3 depthwise launch, backbones/mobilenet_v1.py).  `forward` is the plain
    torch form for CPU eval / export."""

    def __init__(self, kernel_size: int, channels: int, stride: int = 2):
        super().__init__()
        if not isinstance(kernel_size, int):
            if kernel_size[0] != kernel_size[1]:
                raise NotImplementedError("BlurPool2D: square kernels only")
            kernel_size = int(kernel_size[0])
        self.kernel_size, self.stride, self.channels = kernel_size, stride, channels
        row = torch.tensor([float(math.comb(kernel_size - 1, i)) for i in range(kernel_size)], dtype=torch.float32)
        k = row[:, None] * row[None, :]
        self.register_buffer("kernel", k / k.sum())

    def depthwise_weight(self) -> Tensor:
        return self.kernel.repeat((self.channels, 1, 1, 1)).contiguous()

    def forward(self, input: Tensor) -> Tensor:
        return nn.functional.conv2d(input, self.depthwise_weight(), None, stride=self.stride,
                                    padding=(self.kernel_size - 1) // 2, groups=self.channels)


def freeze_norm_stats(m: nn.Module):
    """Put normalisation layers in eval mode and stop training their affine parameters (reference :208-215)."""
    if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d, nn.LayerNorm, nn.InstanceNorm1d, nn.InstanceNorm2d)):
        m.eval()
        for p in m.parameters():
            p.requires_grad = False


def rigid_transformation_25d(r: RotationRepr, t: Tensor, s: Tensor, points: Tensor) -> Tensor:
    """Rotate, scale, and shift x/y only ("2.5 D"; reference :38-56).  points: (..., 68, 3)."""
    moved = r.rotate_points(points) * s[..., None, :]
    return torch.cat((moved[..., :2] + t[..., None, :], moved[..., 2:]), dim=-1)


class DeformableHeadKeypoints(nn.Module):
    """68 keypoints of the BFM head model as a linear basis: keypts + sum_i eig_i * param_i (reference
    :59-82).

    The reference builds the two buffers from facemodel/bfm_noneck_v3.pkl, a large blob that is NOT
    part of the reference checkout (.MISSING_LARGE_BLOBS).  Here they are ordinary state-dict buffers
    (same names/shapes) that arrive with a checkpoint or through `set_basis`; a fresh module holds
    zeros until then.  If the blob is present at FACEMODEL_DIR/bfm_keypoints.npz (arrays `keypts`
    (68,3), `keyeigvecs` (50,68,3)) it is loaded at construction."""

    def __init__(self, num_shape=40, num_expr=10):
        super().__init__()
        self.num_shape, self.num_expr = num_shape, num_expr
        self.num_eigvecs = num_shape + num_expr
        keypts = torch.zeros((68, 3))
        keyeigvecs = torch.zeros((self.num_eigvecs, 68, 3))
        blob = os.path.join(FACEMODEL_DIR, "bfm_keypoints.npz")
        if os.path.exists(blob):
            d = np.load(blob)
            keypts, keyeigvecs = torch.from_numpy(d["keypts"]).float(), torch.from_numpy(d["keyeigvecs"]).float()
        self.register_buffer("keypts", keypts.contiguous())
        self.register_buffer("keyeigvecs", keyeigvecs.contiguous())

    def set_basis(self, keypts, keyeigvecs):
        self.keypts.copy_(torch.as_tensor(keypts))
        self.keyeigvecs.copy_(torch.as_tensor(keyeigvecs))

    def forward(self, shapeparams: Tensor) -> Tensor:
        return torch.einsum("...i,ipd->...pd", shapeparams, self.keyeigvecs) + self.keypts


class PosedDeformableHead(nn.Module):
    def __init__(self, deformable_head: DeformableHeadKeypoints):
        super().__init__()
        self.deformable_head = deformable_head

    def forward(self, coord: Tensor, rots: RotationRepr, params: Tensor) -> Tensor:
        return rigid_transformation_25d(rots, coord[..., :2], coord[..., 2:], self.deformable_head(params))


class LocalToGlobalCoordinateOffset(nn.Module):
    """Per-dataset correction of the predicted pose: rotation about x, translation in the head frame and
    a scale factor, one row of `p` per dataset id (reference :136-184).

    Reference quirk kept bit-for-bit: p[:,1] is used both as the rotation angle and as the first
    translation component, p[:,0] is unused (reference :146-156)."""

    def __init__(self, num_parameter_sets: int = 1):
        super().__init__()
        self.p = nn.Parameter(torch.zeros((num_parameter_sets, 4)))

    def _compute_trafo(self, rot_repr_class: Type[RotationRepr], set_id):
        rows = self.p[:1] if set_id is None else self.p[set_id.long() if isinstance(set_id, Tensor) else set_id]
        quat = rot_repr_class.make_rotate_x(rows[:, 1])
        transl = torch.cat((torch.zeros_like(rows[:, :1]), rows[:, 1:3]), dim=-1)
        return quat, transl, smoothclip0(rows[:, 3])

    def forward(self, quats: RotationRepr, coords: Tensor, set_id: Optional[Tensor]):
        off_q, off_t, off_s = self._compute_trafo(type(quats), set_id)
        scale = coords[..., 2:] * off_s[..., None]
        shift = quats.rotate_points(off_t[..., None, :]).squeeze(-2)[..., :2] * scale
        return quats.mult(off_q), torch.cat((shift + coords[..., :2], scale), dim=-1)


class GaussianMixture(nn.Module):
    """Diagonal-covariance Gaussian mixture log-likelihood (reference :218-290)."""

    def __init__(self, weights: Tensor, means: Tensor, cov: Tensor):
        super().__init__()
        assert weights.shape == means.shape[:1] == cov.shape[:1] and means.shape == cov.shape
        self.cov = cov
        self.register_buffer("weights", weights)
        self.register_buffer("means", means)
        self.register_buffer("scales_inv", cov.rsqrt())
        self.register_buffer("norm_constant", torch.tensor(0.5 * means.shape[-1] * np.log(2 * np.pi), dtype=weights.dtype))

    @property
    def n_components(self) -> int:
        return self.weights.shape[0]

    @staticmethod
    def from_sklearn(gmm) -> "GaussianMixture":
        return GaussianMixture(torch.from_numpy(gmm.weights_), torch.from_numpy(gmm.means_), torch.from_numpy(gmm.covariances_))

    @staticmethod
    def from_npz(filename: str) -> "GaussianMixture":
        d = np.load(filename)
        return GaussianMixture(torch.from_numpy(d["weights"]), torch.from_numpy(d["means"]), torch.from_numpy(d["cov"]))

    @staticmethod
    def from_hdf5(f) -> "GaussianMixture":
        import h5py  # optional dependency, only for the reference's .h5 container

        if isinstance(f, str):
            with h5py.File(f, "r") as file:
                return GaussianMixture.from_hdf5(file)
        assert f.attrs["covariance_type"] == "diag"
        return GaussianMixture(torch.from_numpy(f["weights"][...]), torch.from_numpy(f["means"][...]), torch.from_numpy(f["cov"][...]))

    def save_to_hdf5(self, f, group_name):
        g = f.create_group(group_name) if group_name is not None else f
        g.create_dataset("weights", data=self.weights.cpu().numpy())
        g.create_dataset("means", data=self.means.cpu().numpy())
        g.create_dataset("cov", data=self.cov.cpu().numpy())
        g.attrs["covariance_type"] = "diag"
        return g

    def forward(self, x: Tensor) -> Tensor:
        z = (x[..., None, :] - self.means) * self.scales_inv
        logits = torch.log(self.weights) - 0.5 * z.square().sum(-1) + torch.log(self.scales_inv).sum(-1) - self.norm_constant
        return torch.logsumexp(logits, dim=-1)
