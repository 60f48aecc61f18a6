"""2-D affine transforms in (..., 2, 3) matrix form (reference: neuralnets/affine2d.py): the algebra the
crop / augmentation bookkeeping is written in.  Host-side torch utility; the batched GPU augmentation
(datatransformation/gpu.py) evaluates the same formulas in csrc/warp.hip."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from .math import matvecmul

SQRT2 = torch.sqrt(torch.tensor(2.0))
MaybeTensor = Optional[torch.Tensor]


def _f32(*xs):
    return [torch.as_tensor(x).to(dtype=torch.float32) for x in xs]


def _pack(a, b, tx, c, d, ty) -> torch.Tensor:
    a, b, tx, c, d, ty = torch.broadcast_tensors(*_f32(a, b, tx, c, d, ty))
    return torch.stack((torch.stack((a, b, tx), dim=-1), torch.stack((c, d, ty), dim=-1)), dim=-2)


class Affine2d:
    def __init__(self, m: torch.Tensor):
        assert m.dim() >= 2 and m.shape[-2:] == (2, 3)
        self.m: torch.Tensor = m.to(torch.float32).detach()

    # ---- constructors
    @staticmethod
    def identity(device=None):
        return Affine2d(torch.eye(2, 3, device=device))

    @staticmethod
    def trs(translations: MaybeTensor = None, angles: MaybeTensor = None, scales: MaybeTensor = None):
        """translate(t) . rotate(angle) . scale(s)"""
        ref = translations[..., 0] if translations is not None else (angles if angles is not None else scales)
        one, zero = torch.ones_like(ref, dtype=torch.float32), torch.zeros_like(ref, dtype=torch.float32)
        cs, sn = (one, zero) if angles is None else (torch.cos(angles), torch.sin(angles))
        if scales is not None:
            cs, sn = cs * scales, sn * scales
        tx, ty = (zero, zero) if translations is None else (translations[..., 0], translations[..., 1])
        return Affine2d(_pack(cs, -sn, tx, sn, cs, ty))

    @staticmethod
    def trs_inv(translations: MaybeTensor = None, angles: MaybeTensor = None, scales: MaybeTensor = None):
        """Inverse of trs() with the same arguments."""
        ref = translations[..., 0] if translations is not None else (angles if angles is not None else scales)
        one, zero = torch.ones_like(ref, dtype=torch.float32), torch.zeros_like(ref, dtype=torch.float32)
        cs, sn = (one, zero) if angles is None else (torch.cos(angles), torch.sin(angles))
        if scales is not None:
            cs, sn = cs / scales, sn / scales
        if translations is None:
            tx, ty = zero, zero
        else:
            tx = -(cs * translations[..., 0] + sn * translations[..., 1])
            ty = -(-sn * translations[..., 0] + cs * translations[..., 1])
        return Affine2d(_pack(cs, sn, tx, -sn, cs, ty))

    @staticmethod
    def horizontal_flip(xcenter: torch.Tensor):
        z = torch.zeros_like(xcenter, dtype=torch.float32)
        return Affine2d(_pack(z - 1.0, z, 2 * xcenter, z, z + 1.0, z))

    @staticmethod
    def range_remap(inmin, inmax, outmin, outmax):
        """Same scalar scale on both axes; inputs broadcast to (...)."""
        inmin, inmax, outmin, outmax = _f32(inmin, inmax, outmin, outmax)
        s = (outmax - outmin) / (inmax - inmin)
        off = outmin - inmin * s
        z = torch.zeros_like(s)
        return Affine2d(_pack(s, z, off, z, s, off))

    @staticmethod
    def range_remap_2d(inmin, inmax, outmin, outmax):
        """Per-axis linear map of the box [inmin, inmax] onto [outmin, outmax]; inputs (..., 2)."""
        inmin, inmax, outmin, outmax = _f32(inmin, inmax, outmin, outmax)
        s = (outmax - outmin) / (inmax - inmin)
        off = outmin - inmin * s
        z = torch.zeros_like(s[..., 0])
        return Affine2d(_pack(s[..., 0], z, off[..., 0], z, s[..., 1], off[..., 1]))

    # ---- views
    def tensor(self):
        return self.m

    def tensor33(self):
        bottom = self.m.new_tensor([0.0, 0.0, 1.0]).expand(*self.m.shape[:-2], 1, 3)
        return torch.cat((self.m, bottom), dim=-2)

    def to(self, *args, **kwargs):
        return Affine2d(self.m.to(*args, **kwargs))

    @property
    def R(self):
        return self.m[..., :2, :2]

    @property
    def R33(self):
        r = torch.zeros(*self.m.shape[:-2], 3, 3, dtype=self.m.dtype, device=self.m.device)
        r[..., :2, :2] = self.R
        r[..., 2, 2] = 1.0
        return r

    @property
    def T(self):
        return self.m[..., :2, 2]

    def size(self, i):
        return self.m.size(i)

    @property
    def shape(self):
        return self.m.shape[:-2]

    def __matmul__(self, other: "Affine2d") -> "Affine2d":
        a, b = torch.broadcast_tensors(self.m, other.m)
        rot = a[..., :2, :2] @ b[..., :2, :2]
        t = matvecmul(a[..., :2, :2], b[..., :2, 2]) + a[..., :2, 2]
        return Affine2d(torch.cat((rot, t[..., None]), dim=-1))

    def inv(self) -> "Affine2d":
        r = torch.inverse(self.R)
        return Affine2d(torch.cat((r, -matvecmul(r, self.T)[..., None]), dim=-1))

    @property
    def scales(self):
        """Isotropic scale recovered as |R|_F / sqrt(2)."""
        return torch.linalg.matrix_norm(self.m[..., :, :2]) / SQRT2

    @property
    def det(self):
        m = self.m
        return m[..., 0, 0] * m[..., 1, 1] - m[..., 0, 1] * m[..., 1, 0]

    def __getitem__(self, val):
        return Affine2d(self.m.__getitem__(val))

    def reshape(self, shape):
        return Affine2d(self.m.reshape(tuple(shape) + (2, 3)))

    def expand(self, *shape):
        return Affine2d(self.m.expand(*shape, -1, -1))

    def repeat(self, size: Tuple[int, ...]):
        return Affine2d(self.m.repeat(tuple(size) + (1, 1)))

    def view(self, *shape):
        return Affine2d(self.m.view(*shape, 2, 3))


def roi_normalizing_transform(roi: torch.Tensor) -> Affine2d:
    """Maps each roi (x0,y0,x1,y1) onto [-1,1]^2."""
    assert roi.shape[-1] == 4
    lead = roi.shape[:-1]
    flat = roi.reshape(-1, 4)
    lo = torch.full((flat.shape[0], 2), -1.0, dtype=roi.dtype, device=roi.device)
    return Affine2d.range_remap_2d(flat[:, :2], flat[:, 2:], lo, -lo).reshape(lead)
