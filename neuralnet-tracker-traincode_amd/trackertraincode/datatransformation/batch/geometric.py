"""ROI randomisation of the crop augmentation (reference: datatransformation/batch/geometric.py:29-177).
The per-sample OpenCV image path of the reference's `GeneralFocusRoi.__call__` is replaced by the batched GPU
kernel (datatransformation/gpu.py); the arithmetic that decides WHERE to crop is kept here under the
reference's names, in torch, and is what csrc/warp.hip:view_roi_k reproduces bit-exactly."""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np
import torch
from torch import Tensor

from ...neuralnets.affine2d import Affine2d
from ..tensors.affinetrafo import position_normalization, position_unnormalization


class RoiFocusRandomizationParameters(NamedTuple):
    scales: Tensor        # (B)
    angles: Tensor        # (B)
    translations: Tensor  # (B, 2)
    upfilter: Optional[str] = None
    downfilter: Optional[str] = None


class MakeRoiRandomizationParameters:
    """scale ~ N(extension_factor, 0.1) clipped +-0.5; shift ~ N(0, 0.5) clipped +-1; angle in {0 (2/3), +-a (1/3)}"""

    def __init__(self, rotation_aug_angle, extension_factor):
        self.rotation_aug_angle, self.extension_factor = rotation_aug_angle, extension_factor

    def __call__(self, B: tuple, generator: torch.Generator | None = None, device="cpu") -> RoiFocusRandomizationParameters:
        kw = dict(generator=generator, device=device)
        scales = torch.randn(B, **kw).mul(0.1).clip(-0.5, 0.5).add(self.extension_factor)
        translations = torch.randn(tuple(B) + (2,), **kw).mul(0.5).clip(-1.0, 1.0)
        if self.rotation_aug_angle:
            sign = torch.randint(0, 2, B, **kw).float() * 2.0 - 1.0
            on = (torch.rand(B, **kw) < 1.0 / 3.0).float()
            angles = sign * on * (np.pi * self.rotation_aug_angle / 180.0)
        else:
            angles = torch.zeros(B, device=device)
        return RoiFocusRandomizationParameters(scales, angles, translations, "linear", "area")


class NoRoiRandomization:
    def __init__(self, extent_factor):
        self.extent_factor = extent_factor

    def __call__(self, B, generator=None, device="cpu") -> RoiFocusRandomizationParameters:
        return RoiFocusRandomizationParameters(torch.full(B, self.extent_factor, device=device), torch.zeros(B, device=device),
                                               torch.zeros(tuple(B) + (2,), device=device))


class GeneralFocusRoi:
    def __init__(self, make_randomization_parameters, new_size, roi_variable="roi", insert_backtransform=False):
        self.new_size, self.roi_variable, self.insert_backtransform = new_size, roi_variable, insert_backtransform
        self._max_beyond_border_shift = 0.3
        self.make_randomization_parameters = make_randomization_parameters

    @staticmethod
    def _compute_view_roi(face_bbox: Tensor, enlargement_factor: Tensor, translation_factor: Tensor, beyond_border_shift: float):
        """Square view around the face box: side = max(w,h)*enlargement; shifted by translation_factor in [-1,1]
        times the room the face has inside the view plus `beyond_border_shift` of the smaller of the two
        (reference :108-157)."""
        assert face_bbox.shape[:-1] == enlargement_factor.shape == translation_factor.shape[:-1]
        x0, y0, x1, y1 = face_bbox.unbind(-1)
        rx, ry = translation_factor.unbind(-1)
        w, h = x1 - x0, y1 - y0
        cx, cy = 0.5 * (x1 + x0), 0.5 * (y1 + y0)
        size = torch.maximum(w, h) * enlargement_factor
        tx = (0.5 * torch.abs(size - w) + beyond_border_shift * torch.minimum(size, w)) * rx
        ty = (0.5 * torch.abs(size - h) + beyond_border_shift * torch.minimum(size, h)) * ry
        return torch.stack((cx - size * 0.5 + tx, cy - size * 0.5 + ty, cx + size * 0.5 + tx, cy + size * 0.5 + ty), dim=-1)

    def _compute_point_transform_from_roi(self, B, new_roi: Tensor, new_size: int) -> Affine2d:
        lo, hi = new_roi[..., :2].to(torch.float32), new_roi[..., 2:].to(torch.float32)
        return Affine2d.range_remap_2d(lo, hi, torch.zeros_like(lo), torch.full_like(lo, float(new_size)))

    def _center_rotation_tr(self, rotations: Tensor) -> Affine2d:
        n = self.new_size
        return position_unnormalization(n, n).to(rotations.device) @ Affine2d.trs(angles=rotations) @ position_normalization(n, n).to(rotations.device)

    def transform_for(self, roi: Tensor, params: RoiFocusRandomizationParameters):
        """(integer view roi, Affine2d crop transform) - reference __call__ :198-207."""
        view = torch.round(self._compute_view_roi(roi, params.scales, params.translations, self._max_beyond_border_shift)).to(torch.int32)
        tr = self._center_rotation_tr(params.angles) @ self._compute_point_transform_from_roi(roi.shape[:-1], view, self.new_size)
        return view, tr

    def __call__(self, sample):
        raise NotImplementedError("per-sample CPU cropping (OpenCV) is not part of this package: use datatransformation.gpu.GpuFocusRoiAugment")
