"""Weighted sampling over concatenated datasets (reference: datasets/randomized.py:37-112).

`make_concat_dataset_item_sampler(ConcatDataset, weights)` yields global item indices: for every draw a dataset is
chosen with probability proportional to its weight and that dataset's own sampler (random permutation by default,
restarted when exhausted) supplies the local index.  This is what makes dataset mixing ratios independent of dataset
sizes in the reference's training set (pipelines.py:455-502)."""
from __future__ import annotations

import copy
import sys
from typing import Callable, Iterator, Optional, Sequence

import numpy as np
import torch
from torch.quasirandom import SobolEngine
from torch.utils.data import ConcatDataset, RandomSampler, Sampler

from .. import utils

ChoicesSampler = Callable[[], int]


def weights_normalized(w) -> np.ndarray:
    w = np.asarray(w, dtype=np.float64)
    if w.ndim != 1 or not (w.sum() > 0.0) or (w < 0).any():
        raise ValueError("weights must be a non-negative vector with a positive sum")
    return w / w.sum()


class PseudoRandomChoices:
    """Index i with probability weights[i] (numpy RandomState, reference :59-68)."""

    def __init__(self, weights, seed=None):
        self.probs = weights_normalized(weights)
        self.rng = np.random.RandomState(seed=seed)

    def __call__(self) -> int:
        return int(self.rng.choice(len(self.probs), p=self.probs))

    def draw(self, n: int) -> np.ndarray:
        """n choices at once (the HBM-resident loader draws a whole step)."""
        return self.rng.choice(len(self.probs), size=n, p=self.probs)


class SobolChoices:
    """The same distribution from a scrambled Sobol sequence: dataset shares per batch vary less (reference :46-56)."""

    def __init__(self, weights, seed=None):
        self.accum = torch.cumsum(torch.from_numpy(weights_normalized(weights)), dim=0)
        self.qrng = SobolEngine(1, scramble=True, seed=seed)

    def __call__(self) -> int:
        u = self.qrng.draw().to(self.accum.dtype).reshape(())
        return int(torch.clamp(torch.searchsorted(self.accum, u), 0, len(self.accum) - 1))


class ConcatDatasetSampler(Sampler):
    def __init__(self, dataset: ConcatDataset, wrapped: Sequence[Sampler], dataset_index_sampler: ChoicesSampler,
                 stop_after: int = sys.maxsize):
        self.stop_after, self.samplers, self.dataset_index_sampler = stop_after, list(wrapped), dataset_index_sampler
        self.offsets = [0] + [int(c) for c in dataset.cumulative_sizes[:-1]]

    def __iter__(self) -> Iterator:
        choose = copy.deepcopy(self.dataset_index_sampler)  # every pass over the sampler replays the same dataset sequence
        streams = [utils.cycle(s) for s in self.samplers]
        for _ in range(self.stop_after):
            i = choose()
            item = next(streams[i])
            yield item + self.offsets[i] if isinstance(item, int) else [j + self.offsets[i] for j in item]

    def __len__(self) -> int:
        return self.stop_after


def make_concat_dataset_item_sampler(dataset: ConcatDataset, weights: Sequence[float],
                                     wrapped: Optional[Sequence[Sampler]] = None, stop_after: int = sys.maxsize):
    if wrapped is None:
        wrapped = [RandomSampler(ds) for ds in dataset.datasets]
    if len(wrapped) != len(dataset.datasets) or len(weights) != len(wrapped):
        raise ValueError("one weight and one sampler per concatenated dataset")
    return ConcatDatasetSampler(dataset, wrapped, PseudoRandomChoices(weights), stop_after)
