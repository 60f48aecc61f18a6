"""Rotation containers handed around between model heads and losses (reference: neuralnets/rotrepr.py):
`QuatRepr` (default) and `Mat33Repr` (6D head).  Thin dataclasses over a tensor; sliceable because the
loss plumbing slices predictions per sub-batch (train.py:394-395)."""
from __future__ import annotations

import dataclasses
from typing import Any, Type

import torch
from torch import Tensor

from . import torch6drotation, torchquaternion
from .math import smoothclip0


@dataclasses.dataclass
class QuatRepr:
    value: Tensor

    def rotate_points(self, pts: Tensor) -> Tensor:
        return torchquaternion.rotate(self.value[..., None, :], pts)

    def mult(self, other: "QuatRepr") -> "QuatRepr":
        return QuatRepr(torchquaternion.mult(self.value, other.value))

    @classmethod
    def make_rotate_x(cls: Any, angle: Tensor) -> "QuatRepr":
        half = 0.5 * angle
        zero = torch.zeros_like(half)
        return QuatRepr(torch.stack((torch.sin(half), zero, zero, torch.cos(half)), dim=-1))

    @classmethod
    def from_features(cls: Type["QuatRepr"], z: Tensor) -> tuple["QuatRepr", Tensor]:
        """(normalised quaternion, unnormalised quaternion); the real part is made positive with
        smoothclip0 because q and -q are the same rotation."""
        unnormalized = torch.cat((z[..., :3], smoothclip0(z[..., 3:])), dim=-1)
        return QuatRepr(torchquaternion.normalized(unnormalized)), unnormalized

    def as_quat(self) -> Tensor:
        return self.value

    @property
    def shape(self):
        return self.value.shape[:-1]

    def __getitem__(self, *args):
        return QuatRepr(self.value.__getitem__(*args))


@dataclasses.dataclass
class Mat33Repr:
    value: Tensor

    def rotate_points(self, pts: Tensor) -> Tensor:
        return (self.value @ pts.mT).mT

    def mult(self, other: "Mat33Repr") -> "Mat33Repr":
        return Mat33Repr(self.value @ other.value)

    @classmethod
    def make_rotate_x(cls: Any, angle: Tensor) -> "Mat33Repr":
        sn, cs = torch.sin(angle), torch.cos(angle)
        one, zero = torch.ones_like(angle), torch.zeros_like(angle)
        rows = (one, zero, zero, zero, cs, -sn, zero, sn, cs)
        return Mat33Repr(torch.stack(rows, dim=-1).view(*angle.shape, 3, 3))

    @classmethod
    def from_6drepr_features(cls: Type["Mat33Repr"], z: Tensor) -> "Mat33Repr":
        return Mat33Repr(torch6drotation.tomatrix(z))

    def as_quat(self) -> Tensor:
        if self.value.is_cuda:  # HIP kernel with the hand-derived backward (csrc/loss_math.h: from_matrix)
            from . import _hipops

            return _hipops.MatToQuatFn.apply(self.value.reshape(-1, 3, 3)).view(*self.value.shape[:-2], 4)
        return torchquaternion.from_matrix(self.value)

    @property
    def shape(self):
        return self.value.shape[:-2]

    def __getitem__(self, *args):
        return Mat33Repr(self.value.__getitem__(*args))


RotationRepr = QuatRepr | Mat33Repr
