"""6D rotation representation (Zhou et al. 2020) with the reference's function names
(reference: neuralnets/torch6drotation.py).  Host-side torch utilities; the `--enable-6drot` head is an
optional configuration of the reference (scripts/train_poseestimator.py:415)."""
from __future__ import annotations

import torch
from torch import Tensor


def _as_two_vectors(z: Tensor) -> Tensor:
    return z.view(*z.shape[:-1], 2, 3)


def orthonormality_loss(m: Tensor) -> Tensor:
    """mean((M M^T - I_2)^2) over the 2x2 Gram matrix of the two 3-vectors (reference :20-24)."""
    assert m.shape[-1] == 6
    v = _as_two_vectors(m)
    gram = v @ v.mT
    return (gram - torch.eye(2, device=m.device, dtype=m.dtype)).square().flatten(-2, -1).mean(-1)


def tomatrix(sixdrot: Tensor) -> Tensor:
    """Gram-Schmidt by cross products; rows x, y', z normalised with eps 1e-6; identity where the result
    is not orthonormal to 1e-3 (reference :27-49)."""
    assert sixdrot.shape[-1] == 6
    lead = sixdrot.shape[:-1]
    x, y = _as_two_vectors(sixdrot.reshape(-1, 6)).unbind(-2)
    z = torch.linalg.cross(x, y, dim=-1)
    y = torch.linalg.cross(z, x, dim=-1)
    out = torch.nn.functional.normalize(torch.stack((x, y, z), dim=-2), dim=-1, eps=1e-6)
    eye = torch.eye(3, device=out.device, dtype=sixdrot.dtype)[None]
    bad = (out @ out.mT - eye).flatten(-2).abs().amax(dim=-1)
    out = torch.where(bad[:, None, None] > 1.0e-3, eye, out)
    return out.view(*lead, 3, 3)


def frommatrix(m: Tensor) -> Tensor:
    assert m.shape[-2:] == (3, 3)
    return m[..., :2, :].flatten(-2, -1)


def rotation_distance_loss(a: Tensor, b: Tensor) -> Tensor:
    """0.75 - 0.25 tr(A B^T) (reference :68-72)."""
    assert a.shape[-2:] == (3, 3) and b.shape[-2:] == (3, 3)
    return 0.75 - 0.25 * (a * b).sum(dim=(-2, -1))
