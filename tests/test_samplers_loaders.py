"""Host-side data plumbing: weighted concat sampler and the segmented-collation loader, with the
cases the reference tests (test/test_samplers.py, test/test_batch.py): interleaving ratios, index offsets, restart
of exhausted datasets, per-tag segmentation."""
import numpy as np
import pytest
import torch
from torch.utils.data import ConcatDataset, DataLoader, Dataset, SequentialSampler

from trackertraincode.datasets.batch import Batch, Metadata
from trackertraincode.datasets.randomized import (ConcatDatasetSampler, PseudoRandomChoices, SobolChoices,
                                                  make_concat_dataset_item_sampler, weights_normalized)
from trackertraincode.datatransformation.loader import (PostprocessingLoader, SampleBySampleLoader, SegmentedCollationDataLoader,
                                                         TransformedDataset)


class _Range(Dataset):
    def __init__(self, n, start=0):
        self.n, self.start = n, start

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise IndexError()
        return torch.as_tensor(i + self.start, dtype=torch.int)


def test_interleaved_datasets_equal_weights():
    n1 = n2 = m = bs = 10
    ds = ConcatDataset([_Range(n1), _Range(n2, start=n1)])
    sampler = make_concat_dataset_item_sampler(ds, wrapped=[SequentialSampler(d) for d in ds.datasets], weights=[0.5, 0.5], stop_after=m * bs)
    y = torch.cat(list(DataLoader(ds, batch_size=bs, sampler=sampler, num_workers=0))).numpy()
    assert len(y) == m * bs == len(sampler)
    # sequential samplers restarted when exhausted: within one dataset every item is seen equally often (+-1)
    h = np.bincount(y, minlength=n1 + n2)
    assert np.abs(h[:n1] - h[:n1].mean()).max() <= 1 and np.abs(h[n1:] - h[n1:].mean()).max() <= 1
    assert abs(h[:n1].sum() - m * bs / 2) < 25  # binomial(100, .5): 5 sigma


def test_weighted_ratio_and_offsets_and_replay():
    ds = ConcatDataset([_Range(1000), _Range(10, start=1000), _Range(100, start=1010)])
    sampler = make_concat_dataset_item_sampler(ds, weights=[6.0, 3.0, 1.0], stop_after=20000)
    idx = np.fromiter(iter(sampler), dtype=np.int64)
    share = [(idx < 1000).mean(), ((idx >= 1000) & (idx < 1010)).mean(), (idx >= 1010).mean()]
    assert np.allclose(share, [0.6, 0.3, 0.1], atol=0.015)  # ratios follow the weights, not the dataset sizes
    assert idx.min() >= 0 and idx.max() < 1110
    small = idx[(idx >= 1000) & (idx < 1010)]
    assert np.abs(np.bincount(small - 1000, minlength=10) - len(small) / 10).max() <= 1  # permutation passes, restarted
    assert np.array_equal((idx < 1000), (np.fromiter(iter(sampler), dtype=np.int64) < 1000))  # same dataset sequence every pass
    with pytest.raises(ValueError):
        make_concat_dataset_item_sampler(ds, weights=[1.0, 1.0])
    with pytest.raises(ValueError):
        weights_normalized([0.0, 0.0])


def test_choices_distributions():
    w = [1.0, 2.0, 5.0]
    for cls in (PseudoRandomChoices, SobolChoices):
        c = cls(w, seed=3)
        draws = np.array([c() for _ in range(4000)])
        assert np.allclose(np.bincount(draws, minlength=3) / 4000, np.array(w) / 8, atol=0.03), cls.__name__
    assert np.allclose(np.bincount(PseudoRandomChoices(w, seed=1).draw(8000), minlength=3) / 8000, np.array(w) / 8, atol=0.02)


class _Tagged(Dataset):
    """single-frame samples (no batch dimension) of two tasks with different label sets"""

    def __len__(self):
        return 24

    def __getitem__(self, i):
        tag = "pose" if i % 3 else "lmk"
        data = {"image": torch.full((1, 8, 8), float(i)), "index": torch.tensor(i)}
        if tag == "pose":
            data["pose"] = torch.tensor([0.0, 0.0, 0.0, 1.0])
        return Batch(Metadata(8, 0, tag), data)


def test_segmented_collation_loader():
    seen = []
    loader = SegmentedCollationDataLoader(_Tagged(), batch_size=6, num_workers=0, segmentation_key_getter=lambda b: b.meta.tag,
                                          postprocess=lambda b: (seen.append(b.meta.tag), b)[1])
    assert len(loader) == 4
    steps = list(loader)
    for batches in steps:
        assert {b.meta.tag for b in batches} == {"pose", "lmk"}
        assert sum(b.meta.batchsize for b in batches) == 6
        for b in batches:
            assert b["image"].shape == (b.meta.batchsize, 1, 8, 8)
            assert ("pose" in b) == (b.meta.tag == "pose")
            assert all((int(i) % 3 != 0) == (b.meta.tag == "pose") for i in b["index"])
    assert len(seen) == 8
    assert sum(1 for _ in loader.iter_unrolled()) == 8


def test_transformed_and_sample_loaders():
    ds = TransformedDataset(_Tagged(), lambda b: Batch(b.meta, {**dict(b.items()), "image": b["image"] + 100.0}))
    assert len(ds) == 24 and float(ds[2]["image"][0, 0, 0]) == 102.0 and sum(1 for _ in zip(range(24), ds)) == 24
    items = list(SampleBySampleLoader(ds, num_workers=0, postprocess=lambda b: int(b["index"])))
    assert items == list(range(24))
    pl = PostprocessingLoader(_Range(10), batch_size=5, postprocess=lambda t: t.sum().item())
    assert list(pl) == [10, 35] and len(pl) == 2 and len(pl.dataset) == 10


def test_resident_loader_draw_plan():
    """Sampling plan of the HBM-resident loader (no kernels): dataset shares, permutation passes, tag grouping."""
    from trackertraincode.datasets.resident import ResidentFrames, ResidentLoader

    def frames(n, tag):
        return ResidentFrames(tag, {"image": torch.zeros(n, 1, 4, 4, dtype=torch.uint8), "roi": torch.zeros(n, 4)})

    loader = ResidentLoader([frames(50, "a"), frames(7, "b"), frames(300, "a")], [6.0, 1.0, 3.0], batchsize=200, steps_per_epoch=3, seed=5)
    counts = np.zeros(3)
    seen_b = []
    for _ in range(40):
        plan = loader.draw()
        assert sum(len(i) for _, i in plan) == 200
        for d, idx in plan:
            counts[d] += len(idx)
            assert idx.min() >= 0 and idx.max() < len(loader.datasets[d])
            if d == 1:
                seen_b += idx.tolist()
    assert np.allclose(counts / counts.sum(), [0.6, 0.1, 0.3], atol=0.02)
    full = len(seen_b) // 7 * 7
    assert np.all(np.bincount(seen_b[:full], minlength=7) == full // 7)  # whole permutation passes before repeats
    with pytest.raises(ValueError):
        ResidentFrames("a", {"image": torch.zeros(3, 1, 4, 4, dtype=torch.uint8)})
