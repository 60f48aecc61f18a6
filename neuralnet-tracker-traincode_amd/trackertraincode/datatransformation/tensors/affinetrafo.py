"""How labels move under an Affine2d (reference: datatransformation/tensors/affinetrafo.py:11-148).
Host-side torch versions (any device); the training-time batched form is csrc/warp.hip."""
from __future__ import annotations

import enum

import torch

from ...facemodel.keypoints68 import flip_map
from ...neuralnets.affine2d import Affine2d
from ...neuralnets.math import affinevecmul
from ...neuralnets.torchquaternion import mult


class FieldCategory(str, enum.Enum):
    """What a Batch field is, so transforms know how to treat it (reference: datasets/dshdf5pose.py:21-31)."""
    general = ""
    image = "img"
    quat = "q"
    xys = "xys"
    roi = "roi"
    points = "pts"
    semseg = "seg"


imagelike_categories = [FieldCategory.image, FieldCategory.semseg]


def position_normalization(w: int, h: int) -> Affine2d:
    return Affine2d.range_remap_2d([0.0, 0.0], [w, h], [-1.0, -1.0], [1.0, 1.0])


def position_unnormalization(w: int, h: int) -> Affine2d:
    return Affine2d.range_remap_2d([-1.0, -1.0], [1.0, 1.0], [0.0, 0.0], [w, h])


def _matrix_for(tr: Affine2d, points: torch.Tensor) -> torch.Tensor:
    """tr's matrix reshaped so that it broadcasts over the point dimensions between batch and coordinates."""
    lead = tr.shape
    assert points.shape[: len(lead)] == lead
    return tr.tensor().view(*lead, *([1] * (points.dim() - len(lead) - 1)), 2, 3)


def transform_points(tr: Affine2d, points: torch.Tensor) -> torch.Tensor:
    """x,y affine; z scaled by sqrt(|det|) (never mirrored)."""
    assert points.size(-1) in (2, 3), f"Bad point array shape: {points.shape}"
    xy = affinevecmul(_matrix_for(tr, points), points[..., :2])
    if points.size(-1) == 2:
        return xy
    z = torch.sqrt(torch.abs(tr.det)).view(*tr.shape, *([1] * (points.dim() - len(tr.shape) - 1))) * points[..., 2]
    return torch.cat((xy, z[..., None]), dim=-1)


def transform_keypoints(tr: Affine2d, points: torch.Tensor) -> torch.Tensor:
    """68 landmarks: as points, and left/right partners swap when the transform mirrors."""
    out = transform_points(tr, points)
    det = tr.det
    if det.dim() == 0:
        return out[..., flip_map, :].contiguous() if det < 0.0 else out
    mirrored = det < 0.0
    if torch.any(mirrored):
        out = out.clone()
        out[mirrored] = out[mirrored][..., flip_map, :]
    return out


def transform_roi(tr: Affine2d, roi: torch.Tensor) -> torch.Tensor:
    """Axis-aligned bounding box of the four transformed corners."""
    x0, y0, x1, y1 = roi.unbind(-1)
    corners = torch.stack((torch.stack((x0, y0), -1), torch.stack((x0, y1), -1), torch.stack((x1, y0), -1), torch.stack((x1, y1), -1)), dim=-2)
    moved = transform_points(tr, corners)
    return torch.cat((moved.amin(dim=-2), moved.amax(dim=-2)), dim=-1)


def transform_coord(tr: Affine2d, coord: torch.Tensor) -> torch.Tensor:
    return torch.cat((affinevecmul(tr.tensor(), coord[..., :2]), (tr.scales * coord[..., 2])[..., None]), dim=-1)


def transform_rot(tr: Affine2d, quat: torch.Tensor) -> torch.Tensor:
    """Pre-multiply by the in-plane rotation of `tr` (angle from its y-column so that a pure mirror gives 0),
    reversed under a mirror; a mirror additionally negates the j,k components."""
    m = tr.tensor()
    sg = torch.sign(tr.det)
    alpha = torch.atan2(-m[..., 0, 1], m[..., 1, 1])
    zero = torch.zeros_like(alpha)
    zrot = torch.stack((zero, zero, torch.sin(0.5 * alpha) * sg, torch.cos(0.5 * alpha)), dim=-1).expand_as(quat)
    out = mult(zrot, quat)
    flip = torch.stack((torch.ones_like(sg), sg, sg, torch.ones_like(sg)), dim=-1)
    return out * flip


_BY_CATEGORY = {FieldCategory.xys: transform_coord, FieldCategory.quat: transform_rot, FieldCategory.roi: transform_roi,
                FieldCategory.points: transform_keypoints}


def apply_affine2d(trafo: Affine2d, key: str, value: torch.Tensor, category):
    assert category not in imagelike_categories
    if key == "image_backtransform":  # keep mapping augmented coordinates back to the original image
        return (Affine2d(value) @ trafo.inv()).tensor()
    fn = _BY_CATEGORY.get(FieldCategory(category) if category is not None else FieldCategory.general)
    return value if fn is None else fn(trafo, value)
