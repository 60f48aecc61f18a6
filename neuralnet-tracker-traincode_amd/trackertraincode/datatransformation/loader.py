"""DataLoader wrappers with the reference's names (reference: datatransformation/loader.py:7-116): CPU-side plumbing
around torch's DataLoader for datasets of `Batch` samples.  `SegmentedCollationDataLoader` is the train loader's type:
it yields `list[Batch]`, one per segmentation key (the task tag), each passed through `postprocess` (device transfer,
GPU augmentation, whitening)."""
from __future__ import annotations

from typing import Any, Callable, Iterator

from torch.utils.data import DataLoader, Dataset

from ..datasets.batch import Batch


class TransformedDataset(Dataset):
    def __init__(self, wrapped: Dataset, transform: Callable[[Batch], Batch]):
        super().__init__()
        self.wrapped, self.transform = wrapped, transform

    def __len__(self):
        return len(self.wrapped)

    def __iter__(self) -> Iterator[Batch]:
        return (self.transform(x) for x in self.wrapped)

    def __getitem__(self, key) -> Batch:
        return self.transform(self.wrapped[key])


def _identity(x):
    return x


def _as_list(items):
    return items


class SegmentedCollationDataLoader:
    def __init__(self, dataset: Dataset, *, batch_size: int, num_workers: int, segmentation_key_getter: Callable[[Batch], Any],
                 pin_memory: bool = False, sampler=None, worker_init_fn=None, postprocess: Callable[[Batch], Batch] = _identity):
        self._loader = DataLoader(dataset=dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers,
                                  collate_fn=Batch.Collation(segmentation_key_getter), worker_init_fn=worker_init_fn,
                                  pin_memory=pin_memory)
        self._postprocess = postprocess

    def __iter__(self) -> Iterator[list[Batch]]:
        for items in self._loader:
            assert isinstance(items, list)
            yield [self._postprocess(item) for item in items]

    def iter_unrolled(self) -> Iterator[Batch]:
        for items in self:
            yield from items

    def __len__(self):
        return len(self._loader)


class PostprocessingLoader:
    def __init__(self, *args, **kwargs):
        self._postprocess = kwargs.pop("postprocess", None) or _identity
        self._loader = DataLoader(*args, **kwargs)

    @property
    def dataset(self) -> Dataset:
        return self._loader.dataset

    def __iter__(self):
        return (self._postprocess(items) for items in self._loader)

    def __len__(self):
        return len(self._loader)


class SampleBySampleLoader:
    """Items one at a time, loaded by worker processes in groups of `num_workers` (reference :83-116)."""

    def __init__(self, dataset: Dataset, *, num_workers: int, pin_memory: bool = False, shuffle=False, sampler=None,
                 worker_init_fn=None, postprocess: Callable | None = None):
        self._loader = DataLoader(dataset=dataset, batch_size=max(1, num_workers), sampler=sampler, num_workers=num_workers,
                                  collate_fn=_as_list, worker_init_fn=worker_init_fn, pin_memory=pin_memory, shuffle=shuffle,
                                  drop_last=False)
        self._postprocess = postprocess or _identity

    @property
    def dataset(self) -> Dataset:
        return self._loader.dataset

    def __iter__(self):
        for items in self._loader:
            assert isinstance(items, list)
            yield from (self._postprocess(item) for item in items)

    def __len__(self):
        return len(self._loader.dataset)
