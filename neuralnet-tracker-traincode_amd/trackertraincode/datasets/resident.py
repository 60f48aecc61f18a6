"""Training data resident in HBM.

The reference decodes JPEG frames and warps crops with OpenCV in CPU worker processes, per sample
(datasets/dshdf5pose.py:198-256, datatransformation/batch/geometric.py:193-231, loader.py:24-58); at > 40 k crops/s
per GPU that pipeline starves the device.  With 288 GB of HBM3E the decoded training frames of the reference's
datasets (grey uint8, e.g. 300W-LP's 120 k frames at 192x192 = 4.5 GB) simply live on the GPU:

    ResidentFrames   one dataset: uint8 frames [N,1,Hs,Ws] + per-frame labels, all on the device, with its task Tag
    ResidentLoader   per step: draws (dataset, frame) pairs with the reference's weighted concat rule
                     (datasets/randomized.py: dataset ~ weights, frame ~ random permutation of that dataset), gathers
                     them on the device, groups them by Tag into `list[Batch]` (Batch.Collation's segmentation), and
                     runs the batched HIP crop/warp (+ label bookkeeping) and the fused intensity augmentation.

Decoding HDF5/JPEG into these tensors is a one-off host job outside this package (no h5py here); the loader's
contract downstream is exactly the train loader's (pipelines.py:534-554)."""
from __future__ import annotations

import dataclasses
from typing import Any, Iterator, Sequence

import numpy as np
import torch

from ..datatransformation.gpu import GpuFocusRoiAugment
from ..datatransformation.tensors.affinetrafo import FieldCategory
from .batch import Batch, Metadata
from .randomized import PseudoRandomChoices

_CATEGORIES = {"image": FieldCategory.image, "coord": FieldCategory.xys, "pose": FieldCategory.quat, "roi": FieldCategory.roi,
               "pt3d_68": FieldCategory.points}


@dataclasses.dataclass
class ResidentFrames:
    tag: Any
    fields: dict  # "image": uint8 [N,1,Hs,Ws]; labels [N,...]; all on one device

    def __post_init__(self):
        n = {int(v.shape[0]) for v in self.fields.values()}
        if len(n) != 1 or "image" not in self.fields or "roi" not in self.fields:
            raise ValueError("fields need equal lengths and at least 'image' and 'roi' (the face box the crop is taken around)")
        self.n = n.pop()

    def __len__(self):
        return self.n


class ResidentLoader:
    def __init__(self, datasets: Sequence[ResidentFrames], weights: Sequence[float], batchsize: int, steps_per_epoch: int,
                 seed: int = 0, crop: GpuFocusRoiAugment | None = None, image_augmentations=None):
        if len(datasets) != len(weights):
            raise ValueError("one weight per dataset")
        self.datasets, self.batchsize, self.steps = list(datasets), int(batchsize), int(steps_per_epoch)
        self._choose = PseudoRandomChoices(weights, seed=seed)
        self._rng = np.random.RandomState(seed + 1)
        self._perm = [self._rng.permutation(len(d)) for d in self.datasets]  # one pass over a dataset before any repeat
        self._pos = [0] * len(self.datasets)
        # the crop kernel leaves images in [0,1) when intensity augmentation follows (which whitens), else whitened
        self._augs = list(image_augmentations or [])
        self._crop = crop or GpuFocusRoiAugment(whiten=not self._augs)
        self._gen = torch.Generator().manual_seed(seed + 2)

    def __len__(self):
        return self.steps

    def _next_indices(self, d: int, n: int) -> np.ndarray:
        out = []
        while n > 0:
            if self._pos[d] >= len(self._perm[d]):
                self._perm[d], self._pos[d] = self._rng.permutation(len(self.datasets[d])), 0
            take = min(n, len(self._perm[d]) - self._pos[d])
            out.append(self._perm[d][self._pos[d]:self._pos[d] + take])
            self._pos[d] += take
            n -= take
        return np.concatenate(out)

    def draw(self) -> list[tuple[int, np.ndarray]]:
        """[(dataset index, frame indices)] of one step, datasets in first-seen order of the draw."""
        which = self._choose.draw(self.batchsize)
        order = list(dict.fromkeys(which.tolist()))
        return [(d, self._next_indices(d, int((which == d).sum()))) for d in order]

    def __iter__(self) -> Iterator[list[Batch]]:
        for _ in range(self.steps):
            by_tag: dict[Any, list[Batch]] = {}
            for d, idx in self.draw():
                ds = self.datasets[d]
                dev = ds.fields["image"].device
                sel = torch.from_numpy(idx).to(dev)
                data = {k: v.index_select(0, sel) for k, v in ds.fields.items()}
                meta = Metadata(tuple(data["image"].shape[-2:][::-1]), len(idx), ds.tag, None,
                                {k: c for k, c in _CATEGORIES.items() if k in data})
                by_tag.setdefault(ds.tag, []).append(Batch(meta, data))
            out = []
            for parts in by_tag.values():
                if len({p.meta.image_wh for p in parts}) == 1:
                    b = parts[0] if len(parts) == 1 else Batch.Collation._collate_group(parts)
                    b = self._crop(b, generator=self._gen)
                else:
                    # datasets of one Tag whose frames differ in size (every shard is padded to its OWN largest frame): the source frames
                    # cannot be stacked, the 129 x 129 crops can - crop each dataset's part, collate the crops (the reference crops
                    # per sample before its collation, datatransformation/loader.py:24-58)
                    b = Batch.Collation._collate_group([self._crop(p, generator=self._gen) for p in parts])
                img = b["image"]
                for aug in self._augs:
                    img = aug.apply(img, aug.sample_params(img.shape[0], self._gen))
                b["image"] = img
                out.append(b)
            yield out


class ResidentEvalLoader:
    """The test loader of the reference (pipelines.py:543-552: PostprocessingLoader over the eval-transformed datasets, ONE Batch per
    iteration, no shuffling) over frames resident in HBM: deterministic FocusRoi crop (enlargement 1.1, no shift, no rotation:
    pipelines.py:330-339 stage "eval"), label bookkeeping and whitening on the GPU."""

    def __init__(self, datasets: Sequence[ResidentFrames], batchsize: int, new_size: int = 129, extension_factor: float = 1.1,
                 roi_from_landmarks: bool = False):
        from ..datatransformation.batch.geometric import NoRoiRandomization

        self.datasets, self.batchsize = list(datasets), int(batchsize)
        self._crop = GpuFocusRoiAugment(new_size=new_size, make_params=NoRoiRandomization(extension_factor), whiten=True,
                                        roi_from_landmarks=roi_from_landmarks)

    def __len__(self):
        return sum((len(d) + self.batchsize - 1) // self.batchsize for d in self.datasets)

    def __iter__(self) -> Iterator[Batch]:
        for ds in self.datasets:
            for lo in range(0, len(ds), self.batchsize):
                data = {k: v[lo:lo + self.batchsize] for k, v in ds.fields.items()}
                n = int(data["image"].shape[0])
                meta = Metadata(tuple(data["image"].shape[-2:][::-1]), n, ds.tag, None, {k: c for k, c in _CATEGORIES.items() if k in data})
                yield self._crop(Batch(meta, data))
