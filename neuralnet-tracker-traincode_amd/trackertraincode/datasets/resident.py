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
contract downstream is exactly the train loader's (pipelines.py:534-554).

Data that does not fit: `ResidentFrames.to_host()` keeps a dataset's frames in PINNED HOST memory instead.  The loaders then gather the
drawn frames on the host (into a pinned staging tensor), copy them to the device on a side HIP stream, and - ResidentLoader - do that for
step t+1 on a background thread while step t trains (`prefetch` batches ahead); the main stream waits on the copy's event before the
crop kernel reads the frames.  Same draws, same kernels: for one seed both placements yield bitwise the same batches
(tests/test_resident_host_gpu.py).  One step of 512 frames of 450 x 450 is 104 MB: 2 ms of PCIe 5 x16 beside a 7 ms step."""
from __future__ import annotations

import dataclasses
from typing import Any, Iterator, Sequence

import numpy as np
import torch

from .. import _hip
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

    @property
    def on_host(self) -> bool:
        """Frames in host memory that a GPU consumes (without a GPU the loaders' draw plan can still be exercised on CPU tensors)."""
        return self.fields["image"].device.type == "cpu" and torch.cuda.is_available()

    def to(self, device) -> "ResidentFrames":
        return ResidentFrames(self.tag, {k: v.to(device) for k, v in self.fields.items()})

    def to_host(self) -> "ResidentFrames":
        """The same frames and labels in pinned host memory (for datasets larger than the HBM budget)."""
        pin = torch.cuda.is_available()
        return ResidentFrames(self.tag, {k: (v.cpu().pin_memory() if pin else v.cpu()) for k, v in self.fields.items()})

    def nbytes(self) -> int:
        return sum(v.numel() * v.element_size() for v in self.fields.values())


def _device_of(datasets, device):
    if device is not None:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:  # torch.cuda.set_device / Stream(device=) want the index
            device = torch.device("cuda", torch.cuda.current_device())
        return device
    for d in datasets:
        if not d.on_host:
            return d.fields["image"].device
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class _HostGather:
    """Frames `idx` of a host-resident dataset on the device: host gather into pinned staging tensors, asynchronous copies on a side stream.
    Returns the device tensors and the event the consumer's stream has to wait for."""

    _THREADS = 8          # row ranges of a large gather are copied in parallel (numpy releases the GIL in take)
    _PARALLEL_FROM = 1 << 22  # bytes

    def __init__(self, device):
        from concurrent.futures import ThreadPoolExecutor

        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self._pool = ThreadPoolExecutor(max_workers=self._THREADS, thread_name_prefix="ResidentLoader-gather")

    def _take(self, src: np.ndarray, idx: np.ndarray, dst: np.ndarray):
        # (numpy.take: measured 5 ms for 512 frames of 256 x 256 on one thread where torch.index_select on uint8 takes 160 ms)
        n = len(idx)
        if dst.nbytes < self._PARALLEL_FROM or n < 2 * self._THREADS:
            np.take(src, idx, axis=0, out=dst)
            return
        step = (n + self._THREADS - 1) // self._THREADS
        jobs = [self._pool.submit(np.take, src, idx[lo:lo + step], 0, dst[lo:lo + step]) for lo in range(0, n, step)]
        for j in jobs:
            j.result()

    def __call__(self, fields: dict, idx: torch.Tensor):
        out = {}
        ii = idx.numpy()
        with torch.cuda.stream(self.stream):
            sel = None
            for k, v in fields.items():
                if v.is_cuda:  # a field that already lives on the device beside host frames (e.g. a `dataset_weight` column): plain gather
                    sel = idx.to(v.device) if sel is None else sel
                    out[k] = v.index_select(0, sel)
                    continue
                stage = torch.empty((idx.numel(),) + tuple(v.shape[1:]), dtype=v.dtype, pin_memory=True)  # caching host allocator: recycled once the copy is done
                self._take(v.numpy(), ii, stage.numpy())
                out[k] = stage.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev


class ResidentLoader:
    def __init__(self, datasets: Sequence[ResidentFrames], weights: Sequence[float], batchsize: int, steps_per_epoch: int,
                 seed: int = 0, crop: GpuFocusRoiAugment | None = None, image_augmentations=None, device=None, prefetch: int = 2):
        if len(datasets) != len(weights):
            raise ValueError("one weight per dataset")
        self._device = _device_of(datasets, device)
        self._host = _HostGather(self._device) if any(d.on_host for d in datasets) else None
        self._prefetch = max(1, int(prefetch))
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

    def _gather_step(self):
        """The drawn frames of one step on the device, grouped by Tag, + the events of their host-to-device copies."""
        by_tag: dict[Any, list[Batch]] = {}
        events = []
        for d, idx in self.draw():
            ds = self.datasets[d]
            if ds.on_host:
                data, ev = self._host(ds.fields, torch.from_numpy(idx))
                events.append(ev)
            else:
                sel = torch.from_numpy(idx).to(ds.fields["image"].device)
                data = {k: v.index_select(0, sel) for k, v in ds.fields.items()}
            meta = Metadata(tuple(data["image"].shape[-2:][::-1]), len(idx), ds.tag, None,
                            {k: c for k, c in _CATEGORIES.items() if k in data})
            by_tag.setdefault(ds.tag, []).append(Batch(meta, data))
        return by_tag, events

    def _gathered_steps(self):
        """`_gather_step()` per step; with host-resident frames from a background thread that runs `prefetch` steps ahead (the draws stay in
        order: one producer)."""
        if self._host is None:
            for _ in range(self.steps):
                yield self._gather_step()
            return
        import queue
        import threading

        q: queue.Queue = queue.Queue(maxsize=self._prefetch)
        stop = threading.Event()

        def produce():
            try:
                torch.cuda.set_device(self._device)
                for _ in range(self.steps):
                    with _hip.CAPTURE_LOCK:  # never allocate / copy while the training thread captures a hipGraph (train.GraphedTrainStep)
                        item = self._gather_step()
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                    if stop.is_set():
                        return
                q.put(None)
            except BaseException as e:  # surfaces in the consumer
                q.put(e)

        t = threading.Thread(target=produce, name="ResidentLoader-prefetch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            t.join(timeout=5.0)

    def __iter__(self) -> Iterator[list[Batch]]:
        for by_tag, events in self._gathered_steps():
            if events:
                cur = torch.cuda.current_stream(self._device)
                for ev in events:
                    cur.wait_event(ev)
                for parts in by_tag.values():  # allocated on the copy stream, read on this one: the allocator must not recycle them under it
                    for part in parts:
                        for v in part.values():
                            if torch.is_tensor(v) and v.is_cuda:
                                v.record_stream(cur)
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
                 roi_from_landmarks: bool = False, device=None):
        from ..datatransformation.batch.geometric import NoRoiRandomization

        self.datasets, self.batchsize = list(datasets), int(batchsize)
        self._device = _device_of(datasets, device)
        self._crop = GpuFocusRoiAugment(new_size=new_size, make_params=NoRoiRandomization(extension_factor), whiten=True,
                                        roi_from_landmarks=roi_from_landmarks)

    def __len__(self):
        return sum((len(d) + self.batchsize - 1) // self.batchsize for d in self.datasets)

    def __iter__(self) -> Iterator[Batch]:
        for ds in self.datasets:
            for lo in range(0, len(ds), self.batchsize):
                data = {k: v[lo:lo + self.batchsize] for k, v in ds.fields.items()}
                if ds.on_host:  # pinned host frames: plain copies on the current stream
                    data = {k: v.to(self._device, non_blocking=True) for k, v in data.items()}
                n = int(data["image"].shape[0])
                meta = Metadata(tuple(data["image"].shape[-2:][::-1]), n, ds.tag, None, {k: c for k, c in _CATEGORIES.items() if k in data})
                yield self._crop(Batch(meta, data))
