"""Small tensor helpers with the reference's names (reference: neuralnets/math.py).

Host-side utilities (parameter initialisation, evaluation scripts); the training arithmetic lives in
the HIP kernels (csrc/head_math.h restates smoothclip0 as `elu1`)."""
from __future__ import annotations

import functools

import torch
import torch.nn.functional as F


def matvecmul(m: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """matrix @ vector for batched operands (reference :8-14)."""
    return (m @ v.unsqueeze(-1)).squeeze(-1)


def affinevecmul(m: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Apply [A|t] stored as (..., n, n+1) to v (reference :17-20)."""
    return matvecmul(m[..., :, :-1], v) + m[..., :, -1]


def random_uniform(shape, minval, maxval, *args, **kwargs):
    return minval + (maxval - minval) * torch.rand(shape, *args, **kwargs)


def random_choice(shape: tuple, values: torch.Tensor, weights: torch.Tensor, replacement):
    count = 1
    for s in shape:
        count *= int(s)
    picks = values[torch.multinomial(weights, count, replacement=replacement)]
    return picks.reshape(shape)


def smoothclip0(x: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """elu(x) + 1: smooth, strictly positive (reference :34-37)."""
    return F.elu(x, inplace=inplace).add_(1.0) if inplace else F.elu(x) + 1.0


def inv_smoothclip0(x) -> torch.Tensor:
    """Inverse of smoothclip0: y-1 for y>1, log y otherwise (reference :40-48)."""
    x = torch.as_tensor(x)
    flat = torch.atleast_1d(x).clone()
    big = flat > 1.0
    out = torch.where(big, flat - 1.0, torch.log(torch.where(big, torch.ones_like(flat), flat)))
    return out.view(x.shape)


def sqrclip0(x: torch.Tensor, beta: float):
    z = F.relu(x + 0.5 * beta)
    return torch.where(z < beta, z.square() * (0.5 / beta), z - 0.5 * beta)


def inv_sqrclip0(y: torch.Tensor, beta: float):
    return torch.where(y > 0.5 * beta, y + 0.5 * beta, torch.sqrt(2.0 * beta * y)) - 0.5 * beta


def chain_gmm(*matrices):
    return functools.reduce(torch.matmul, matrices)
