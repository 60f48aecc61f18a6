"""`Batch` / `Metadata`: the container the loss plumbing consumes (reference: datasets/batch.py).

A Batch is a dict of equally-long tensors plus metadata (task tag, batch size, optional video
sequence offsets, per-field categories).  `Batch.collate` groups samples by tag ("segmented
collation"): the training step receives `list[Batch]`, one per tag (reference :167-236).
"""
from __future__ import annotations

import copy
import dataclasses
from collections import defaultdict
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple, Union

import numpy as np
import torch

TensorOrArray = Union[torch.Tensor, np.ndarray]
Tag = Any


@dataclasses.dataclass
class Metadata:
    _imagesize: Union[int, Tuple[int, int]]
    batchsize: int
    tag: Optional[Any] = None
    seq: Optional[List[int]] = None
    categories: Dict[str, Any] = dataclasses.field(default_factory=dict)

    @property
    def image_wh(self):
        s = self._imagesize
        return s if isinstance(s, tuple) else (s, s)

    @property
    def imagesize(self):
        assert isinstance(self._imagesize, int)
        return self._imagesize

    @property
    def sequence_start_end(self):
        assert self.seq
        return list(zip(self.seq[:-1], self.seq[1:]))

    @property
    def prefixshape(self):
        if self.seq:
            return (self.seq[-1],)
        return (self.batchsize,) if self.batchsize else ()

    @property
    def is_single_frame(self):
        return self.seq is None and self.batchsize == 0


def _concat(items):
    first = items[0]
    return torch.cat(list(items), dim=0) if isinstance(first, torch.Tensor) else np.concatenate(list(items), axis=0)


class Batch:
    def __init__(self, meta: Metadata, *data, **kwargs):
        self.meta: Metadata = meta
        self._data: dict[str, TensorOrArray] = dict(*data, **kwargs)

    @staticmethod
    def from_data_with_categories(meta: Metadata, *args, **kwargs):
        """Values are (tensor, category) pairs; categories go to the metadata."""
        pairs = dict(*args, **kwargs)
        meta = copy.copy(meta)
        meta.categories = dict(meta.categories)
        meta.categories.update((k, c) for k, (_, c) in pairs.items())
        return Batch(meta, ((k, v) for k, (v, _) in pairs.items()))

    # ---- mapping protocol
    def items(self):
        return self._data.items()

    def keys(self):
        return self._data.keys()

    def values(self):
        return self._data.values()

    def __getitem__(self, k):
        return self._data[k]

    def __setitem__(self, k, v):
        self._data[k] = v

    def __delitem__(self, k):
        del self._data[k]

    def __contains__(self, k):
        return k in self._data

    def pop(self, k):
        return self._data.pop(k)

    @property
    def device(self):
        return next(iter(self.values())).device

    def __str__(self):
        seq = f",N={self.meta.seq[-1]}" if self.meta.seq is not None else ""
        return f"Batch({self.meta.tag},B={self.meta.batchsize}{seq})"

    def get_category(self, k, default=None):
        assert k in self._data
        return self.meta.categories.get(k, default)

    def with_batchdim(self) -> "Batch":
        if self.meta.batchsize > 0:
            return self
        meta = copy.copy(self.meta)
        meta.batchsize = 1
        if self.meta.seq is not None:
            return Batch(meta, self.items())
        return Batch(meta, ((k, v[None, ...]) for k, v in self.items()))

    def iter_frames(self) -> Iterator["Batch"]:
        if self.meta.is_single_frame:
            yield self
            return
        (n,) = self.meta.prefixshape
        meta = copy.copy(self.meta)
        meta.batchsize, meta.seq = 0, None
        for i in range(n):
            yield Batch(meta, ((k, v[i, ...]) for k, v in self.items()))

    def iter_sequences(self) -> Iterator["Batch"]:
        assert self.meta.seq is not None
        for a, b in self.meta.sequence_start_end:
            meta = copy.copy(self.meta)
            meta.batchsize, meta.seq = 0, [0, b - a]
            yield Batch(meta, ((k, v[a:b, ...]) for k, v in self.items()))

    def undo_collate(self) -> Iterator["Batch"]:
        if self.meta.seq:
            yield from self.iter_sequences()
        else:
            yield from self.iter_frames()

    def pin_memory(self):
        return Batch(self.meta, ((k, v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in self.items()))

    def copy(self):
        return Batch(self.meta, **self._data)

    def to(self, *args, **kwargs):
        assert all(isinstance(x, torch.Tensor) for x in self._data.values()), "Only applicable to PyTorch"
        return Batch(self.meta, ((k, v.to(*args, **kwargs)) for k, v in self.items()))

    # ---- collation
    class Collation:
        """Callable for DataLoader(collate_fn=...).  With a key getter the samples are split by key and
        a list of batches is returned (one per key, first-seen order), otherwise a single batch.
        Stills (single frames or already-batched) are concatenated along the batch dimension; videos
        are concatenated frame-wise and their sequence offsets shifted (reference :166-236)."""

        def __init__(self, key_getter: Callable[["Batch"], Any] | None = None):
            self._key_getter = key_getter if key_getter is not None else (lambda b: True)
            self._divide_samples = key_getter is not None

        def __call__(self, samples: List["Batch"]):
            groups: dict[Any, list[Batch]] = defaultdict(list)
            for s in samples:
                assert isinstance(s, Batch), f"Expected list of Batch types. Got {type(s)}"
                groups[self._key_getter(s)].append(s)
            out = [self._collate_group(g) for g in groups.values()]
            if not self._divide_samples:
                (out,) = out
            return out

        @staticmethod
        def _collate_group(samples: List["Batch"]) -> "Batch":
            first = samples[0]
            meta = copy.copy(first.meta)
            if first.meta.seq is None:
                meta.batchsize = sum(max(s.meta.batchsize, 1) for s in samples)
                samples = [s.with_batchdim() for s in samples]
            else:
                seq, offset = [0], 0
                for s in samples:
                    seq += [int(e) + offset for e in s.meta.seq[1:]]
                    offset += int(s.meta.seq[-1])
                meta.seq, meta.batchsize = seq, len(seq) - 1
            assert all(s.meta.prefixshape != () for s in samples)
            return Batch(meta, {k: _concat([s[k] for s in samples]) for k in first.keys()})

    collate = Collation()
