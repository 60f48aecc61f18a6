"""On-GPU intensity augmentation with the reference's vocabulary (reference: datatransformation/batch/intensity.py and
the kornia classes it re-exports; call site pipelines.py:508-532).

The reference stacks kornia modules: every augmentation is a pass over the batch.  Here the `Random*` classes only
describe an operation and SAMPLE its per-image parameters; `KorniaImageDistortions` merges what all of them drew into
one parameter table and a single HIP kernel (csrc/intensity.hip) applies the whole chain with the image resident in LDS.

Semantics restated from kornia (not installed here - PARITY UNPINNED, see oracle/intensity.py):
  * every operation fires per sample with probability p and draws its magnitude uniformly from its range;
  * `random_apply=k` picks k of the container's operations per call (uniformly, without replacement) and keeps their
    declared order;
  * operations always run in the fixed order equalize, posterize, gamma, contrast, brightness, blur, noise, clip - the
    order in which the reference declares them.  A container that lists them in another order is rejected;
  * several `RandomGaussianNoise` rungs that fire for the same image add independent normal draws; their sum is one
    normal draw with the root-sum-square std, which is what the kernel applies (one noise tensor per call).
"""
from __future__ import annotations

from copy import copy

import torch

from ... import _hip
from ...datasets.batch import Batch

_EQ, _POST, _GAMMA, _CONTRAST, _BRIGHT, _BLUR, _NOISE, _NPARAMS = 0, 1, 2, 3, 4, 5, 6, 8


class _Op:
    slot = -1
    order = -1

    def __init__(self, p: float):
        self.p = float(p)

    def fires(self, n, g):
        return torch.rand(n, generator=g) < self.p

    def sample(self, n, g) -> torch.Tensor:  # value written to the parameter slot where the operation fires
        return torch.ones(n)


class _RangeOp(_Op):
    def __init__(self, rng, p):
        super().__init__(p)
        self.lo, self.hi = (float(rng[0]), float(rng[1])) if isinstance(rng, (tuple, list)) else (float(rng), float(rng))

    def sample(self, n, g):
        return self.lo + (self.hi - self.lo) * torch.rand(n, generator=g)


class RandomEqualize(_Op):
    slot, order = _EQ, 0

    def __init__(self, p=0.5):
        super().__init__(p)


class RandomPosterize(_RangeOp):
    """bits ~ U(lo, hi) truncated to an integer (kornia: `.int()`), 8 = identity."""
    slot, order = _POST, 1

    def __init__(self, bits=3, p=0.5):
        super().__init__(bits, p)

    def sample(self, n, g):
        return super().sample(n, g).floor().clamp(0, 8)


class RandomGamma(_RangeOp):
    slot, order = _GAMMA, 2

    def __init__(self, gamma=(1.0, 1.0), gain=(1.0, 1.0), p=0.5):
        if tuple(gain) != (1.0, 1.0):
            raise NotImplementedError("gain != 1 is not used by the reference (pipelines.py:512)")
        super().__init__(gamma, p)


class RandomContrast(_RangeOp):
    slot, order = _CONTRAST, 3

    def __init__(self, contrast=(1.0, 1.0), p=0.5):
        super().__init__(contrast, p)


class RandomBrightness(_RangeOp):
    slot, order = _BRIGHT, 4

    def __init__(self, brightness=(1.0, 1.0), p=0.5):
        super().__init__(brightness, p)


class RandomGaussianBlur(_Op):
    slot, order = _BLUR, 5

    def __init__(self, kernel_size=(5, 5), sigma=(1.5, 1.5), p=0.5, border_type="reflect", silence_instantiation_warning=True):
        if tuple(kernel_size) != (5, 5) or tuple(float(s) for s in sigma) != (1.5, 1.5) or border_type != "reflect":
            raise NotImplementedError("the kernel is built for the reference's blur: 5x5, sigma 1.5, reflect border (pipelines.py:515-517)")
        super().__init__(p)


class RandomGaussianNoise(_Op):
    slot, order = _NOISE, 6

    def __init__(self, mean=0.0, std=1.0, p=0.5):
        if mean != 0.0:
            raise NotImplementedError("mean != 0 is not used by the reference")
        super().__init__(p)
        self.std = float(std)

    def sample(self, n, g):
        return torch.full((n,), self.std)


class OnlyClip(_Op):
    """clip(0, 1): the kernel always ends with it (reference :55-64)."""
    slot, order = -1, 7

    def __init__(self, p=1.0):
        super().__init__(p)


RandomGaussianNoiseWithClipping = RandomGaussianNoise  # the kernel clips after the noise in any case (reference :43-52)


class KorniaImageDistortions:
    """`KorniaImageDistortions(*ops, random_apply=None)(batch)`: augments every image field of the batch (reference
    :30-41).  `out_shift` folds the whitening that follows in the pipeline (`whiten_batch`, -0.5) into the same pass."""

    def __init__(self, *ops, random_apply=None, out_shift: float = 0.0, generator: torch.Generator | None = None):
        orders = [o.order for o in ops]
        if any(not isinstance(o, _Op) for o in ops) or orders != sorted(orders):
            raise NotImplementedError("operations must be this module's Random* classes in the kernel's fixed order "
                                      "(equalize, posterize, gamma, contrast, brightness, blur, noise, clip)")
        self.ops, self.random_apply, self.out_shift, self.generator = list(ops), random_apply, float(out_shift), generator

    def sample_params(self, n: int, generator: torch.Generator | None = None) -> torch.Tensor:
        """[n, 8] float32 parameter table on the CPU (include/ttk.h TTK_INTENSITY_*)."""
        g = generator if generator is not None else self.generator
        ops = self.ops
        if self.random_apply is not None and self.random_apply < len(ops):
            keep = torch.randperm(len(ops), generator=g)[: int(self.random_apply)].sort().values.tolist()
            ops = [ops[i] for i in keep]
        prm = torch.zeros((n, _NPARAMS), dtype=torch.float32)
        for op in ops:
            fire = op.fires(n, g)
            val = op.sample(n, g).to(torch.float32)
            if op.slot < 0:
                continue
            if op.slot == _NOISE:  # independent rungs: variances add
                prm[:, _NOISE] = torch.sqrt(prm[:, _NOISE] ** 2 + torch.where(fire, val, torch.zeros_like(val)) ** 2)
            else:
                prm[:, op.slot] = torch.where(fire, val, prm[:, op.slot])
        return prm

    def apply(self, image: torch.Tensor, params: torch.Tensor, noise: torch.Tensor | None = None) -> torch.Tensor:
        """image [B,1,H,W] (or [B,H,W]) float32 in [0,1] on the GPU; params from sample_params."""
        if not image.is_cuda:
            raise RuntimeError("KorniaImageDistortions runs in a HIP kernel: CUDA tensors required (no CPU fallback)")
        x = image.to(torch.float32).contiguous()
        if x.dim() == 4 and x.shape[1] != 1:
            raise NotImplementedError("grey-level crops only (one channel), as in the reference's pipeline")
        B, H, W = x.shape[0], x.shape[-2], x.shape[-1]
        prm = params.to(device=x.device, dtype=torch.float32).contiguous()
        if noise is None and bool((params[:, _NOISE] > 0).any()):
            noise = torch.randn((B, H, W), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        _hip.lib().call("ttk_intensity_augment", _hip.ptr(x), _hip.ptr(y), _hip.ptr(prm), _hip.ptr(noise), B, H, W, self.out_shift)
        return y

    def __call__(self, batch: Batch) -> Batch:
        from ..tensors.affinetrafo import FieldCategory

        batch = copy(batch)
        for k, v in list(batch.items()):
            if batch.get_category(k) != FieldCategory.image:
                continue
            batch[k] = self.apply(v, self.sample_params(v.shape[0]))
        return batch
