"""Quaternion algebra with the reference's function names (reference: neuralnets/torchquaternion.py).

Component order (i, j, k, w): real part LAST, as scipy.  These are general-purpose torch utilities
(evaluation, label transforms, tests against scipy); inside the training step the same formulas run
in csrc/head_math.h / loss_math.h.  Written component-wise instead of the reference's 4x4 matrix
form, same values to fp32 rounding.
"""
from __future__ import annotations

from typing import Final, Union

import numpy as np
import torch
from torch import Tensor

iw: Final[int] = 3
ii: Final[int] = 0
ij: Final[int] = 1
ik: Final[int] = 2
iijk: Final[slice] = slice(0, 3)


def mult(u: Tensor, v: Tensor) -> Tensor:
    """Hamilton product u*v (reference :40-48)."""
    ui, uj, uk, uw = u.unbind(-1)
    vi, vj, vk, vw = v.unbind(-1)
    return torch.stack(
        (
            uw * vi + ui * vw + uj * vk - uk * vj,
            uw * vj - ui * vk + uj * vw + uk * vi,
            uw * vk + ui * vj - uj * vi + uk * vw,
            uw * vw - ui * vi - uj * vj - uk * vk,
        ),
        dim=-1,
    )


def conjugate(q: Tensor) -> Tensor:
    return torch.cat((-q[..., :3], q[..., 3:]), dim=-1)


def rotate(q: Tensor, p: Tensor) -> Tensor:
    """q (p,0) q^* (reference :51-67); broadcasting over leading dimensions."""
    lead = torch.broadcast_shapes(q.shape[:-1], p.shape[:-1])
    q, p = q.expand(*lead, 4), p.expand(*lead, 3)
    pq = torch.cat((p, torch.zeros_like(p[..., :1])), dim=-1)
    return mult(mult(q, pq), conjugate(q))[..., :3]


def tomatrix(q: Tensor) -> Tensor:
    """Rotation matrix of a unit quaternion (reference :70-91)."""
    i, j, k, w = q.unbind(-1)
    m = torch.stack(
        (
            1 - 2 * (j * j + k * k), 2 * (i * j - k * w), 2 * (i * k + j * w),
            2 * (i * j + k * w), 1 - 2 * (i * i + k * k), 2 * (j * k - i * w),
            2 * (i * k - j * w), 2 * (j * k + i * w), 1 - 2 * (i * i + j * j),
        ),
        dim=-1,
    )
    return m.view(*q.shape[:-1], 3, 3)


def positivereal(q: Tensor) -> Tensor:
    return q * torch.sign(q[..., iw:])


def from_matrix(m: Tensor) -> Tensor:
    """Best-conditioned of the four closed forms (reference :94-168): pick the largest of
    1+-m00+-m11+-m22, divide the off-diagonal combinations by it."""
    assert m.shape[-2:] == (3, 3)
    lead = m.shape[:-2]
    m = m.reshape(-1, 3, 3)
    m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    args = torch.stack((1 - m00 - m11 + m22, 1 - m00 + m11 - m22, 1 + m00 - m11 - m22, 1 + m00 + m11 + m22), dim=-1)
    args = args.clamp_min(1.0e-6)
    r = 0.5 * torch.sqrt(args)  # qk, qj, qi, qw candidates from their own diagonal combination
    a10, a01 = m[:, 1, 0], m[:, 0, 1]
    a20, a02 = m[:, 2, 0], m[:, 0, 2]
    a21, a12 = m[:, 2, 1], m[:, 1, 2]
    q4 = 0.25
    from_k = torch.stack((q4 * (a20 + a02) / r[:, 0], q4 * (a12 + a21) / r[:, 0], r[:, 0], q4 * (a10 - a01) / r[:, 0]), -1)
    from_j = torch.stack((q4 * (a10 + a01) / r[:, 1], r[:, 1], q4 * (a21 + a12) / r[:, 1], q4 * (a02 - a20) / r[:, 1]), -1)
    from_i = torch.stack((r[:, 2], q4 * (a10 + a01) / r[:, 2], q4 * (a02 + a20) / r[:, 2], q4 * (a21 - a12) / r[:, 2]), -1)
    from_w = torch.stack((q4 * (a21 - a12) / r[:, 3], q4 * (a02 - a20) / r[:, 3], q4 * (a10 - a01) / r[:, 3], r[:, 3]), -1)
    cands = torch.stack((from_k, from_j, from_i, from_w), dim=1)
    with torch.no_grad():
        pick = torch.argmax(args, dim=-1)
    q = cands[torch.arange(cands.shape[0], device=m.device), pick]
    return positivereal(q).view(*lead, 4)


def from_rotvec(r: Tensor, eps=1.0e-12) -> Tensor:
    angle = torch.linalg.vector_norm(r, dim=-1, keepdim=True)
    axis = r / (angle + eps)
    return torch.cat((axis * torch.sin(0.5 * angle), torch.cos(0.5 * angle)), dim=-1)


def to_rotvec(q: Tensor, eps=1.0e-12) -> Tensor:
    """Axis * angle with angle in [0, pi] (reference :187-199)."""
    q = positivereal(q)
    v = q[..., iijk]
    n = torch.linalg.vector_norm(v, dim=-1, keepdim=True)
    angle = 2.0 * torch.atan2(n, q[..., iw:])
    return v * angle / (n + eps)


def rotation_delta(from_: Tensor, to_: Tensor) -> Tensor:
    return to_rotvec(mult(conjugate(from_), to_))


def slerp(p: Tensor, q: Tensor, t: Union[float, Tensor], eps=1.0e-12) -> Tensor:
    return mult(p, from_rotvec(rotation_delta(p, q) * t))


def normalized(q: Tensor) -> Tensor:
    return torch.nn.functional.normalize(q, p=2.0, dim=-1, eps=1.0e-6)


def distance(a: Tensor, b: Tensor) -> Tensor:
    return 1.0 - (a * b).sum(dim=-1).square()


def geodesicdistance(a: Tensor, b: Tensor) -> Tensor:
    return torch.linalg.vector_norm(rotation_delta(a, b), dim=-1)


def quat_average(quats):
    """Sign-aligned mean of an ensemble [E, N, 4] (numpy; reference :239-256)."""
    quats = np.array(quats, dtype=np.float64)
    assert quats.ndim == 3 and quats.shape[-1] == 4
    pivot = np.argmax(np.abs(quats).sum(axis=0), axis=-1)
    flip = np.take_along_axis(quats, pivot[None, :, None], axis=-1)[..., 0] < 0.0
    quats[flip] *= -1.0
    mean = quats.mean(axis=0)
    norms = np.linalg.norm(mean, axis=-1, keepdims=True)
    if not np.all(norms > 0.5):
        print("quat_average: rotation predictions differ wildly")
    return mean / norms
