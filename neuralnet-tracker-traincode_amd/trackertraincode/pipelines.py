"""Task tags, dataset ids and the loader entry point (reference: trackertraincode/pipelines.py).

In scope for the MI355X path: the `Tag` / `Id` vocabularies and the contract of
`make_pose_estimation_loaders` (train loader yields `list[Batch]`, one Batch per Tag, images already
on the device, f32 [n,1,129,129] in about [-0.5, 0.5]).  The reference's dataset constructors read
unpublished / multi-GB HDF5 files through h5py + OpenCV in worker processes (pipelines.py:399-500) and
are replaced by .npz shards decoded once into HBM-resident frames (datasets/shards.py, datasets/resident.py);
`SyntheticPoseLoader` provides the same contract from seeded tensors for benchmarks and tests.
"""
from __future__ import annotations

import enum
from typing import Iterator, Sequence

import numpy as np
import torch

from . import utils
from .datasets.batch import Batch, Metadata


class Tag(enum.Enum):
    POSE_WITH_LANDMARKS = 1
    SELF_SUPERVISED_POSE = 2
    FACE_DETECTION = 3
    ONLY_LANDMARKS = 4
    ONLY_LANDMARKS_25D = 5
    ONLY_POSE = 7
    POSE_WITH_LANDMARKS_3D_AND_2D = 8
    ONLY_LANDMARKS_2D = 9
    SEMSEG = 10
    POSE_WITH_LMKS_NO_SHAPE_PARAMS = 11


class Id(enum.Enum):
    _300WLP = 2
    SYNFACE = 5
    WFLW_RELABEL = 6
    AFLW2k3d = 8
    BIWI = 9
    WIDER = 11
    _300VW = 12
    LAPA = 13
    REPO_300WLP = 15
    WFLW_LP = 16
    LAPA_MEGAFACE_LP = 17
    REPO_300WLP_WO_EXTRA = 18
    PANOPTIC_CMU = 19
    REPLICANT_FACE = 20


_FIELDS_BY_TAG = {
    Tag.POSE_WITH_LANDMARKS: ("pose", "coord", "roi", "pt3d_68", "shapeparam"),
    Tag.POSE_WITH_LANDMARKS_3D_AND_2D: ("pose", "coord", "roi", "pt3d_68", "shapeparam"),
    Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS: ("pose", "coord", "roi", "pt3d_68"),
    Tag.ONLY_POSE: ("pose", "coord", "roi"),
    Tag.ONLY_LANDMARKS: ("pt3d_68",),
    Tag.ONLY_LANDMARKS_25D: ("pt3d_68",),
}


def synthetic_subbatch(tag: Tag, n: int, device, generator: torch.Generator, inputsize: int = 129) -> Batch:
    """One sub-batch with the label distributions of SURVEY.md §8(d)."""
    g = generator
    r = lambda *s: torch.rand(*s, generator=g, device=g.device)
    rn = lambda *s: torch.randn(*s, generator=g, device=g.device)
    data = {
        "image": r(n, 1, inputsize, inputsize) - 0.5,
        "coord_convention_id": torch.randint(0, 8, (n,), generator=g, device=g.device, dtype=torch.int32),
    }
    fields = _FIELDS_BY_TAG[tag]
    if "pose" in fields:
        data["pose"] = torch.nn.functional.normalize(rn(n, 4), dim=-1)
    if "coord" in fields:
        data["coord"] = torch.stack((r(n) * 0.6 - 0.3, r(n) * 0.6 - 0.3, r(n) * 0.8 + 0.8), dim=-1)
    if "roi" in fields:
        data["roi"] = torch.tensor([-0.85, -0.85, 0.85, 0.85], device=g.device) + (r(n, 4) * 0.2 - 0.1)
    if "pt3d_68" in fields:
        data["pt3d_68"] = rn(n, 68, 3) * 0.5
    if "shapeparam" in fields:
        data["shapeparam"] = rn(n, 50) * 0.5
    return Batch(Metadata(inputsize, batchsize=n, tag=tag), {k: v.to(device) for k, v in data.items()})


def make_image_augmentations(generator: torch.Generator | None = None, whiten: bool = True):
    """The reference's two intensity-augmentation containers (pipelines.py:508-528) on the fused HIP kernel
    (datatransformation/batch/intensity.py); the second one also applies the whitening that follows (:530, -0.5).
    Input images are in [0,1]."""
    from .datatransformation import batch as B

    return [
        B.KorniaImageDistortions(
            B.RandomEqualize(p=0.2), B.RandomPosterize((4.0, 6.0), p=0.01), B.RandomGamma((0.5, 2.0), p=0.2),
            B.RandomContrast((0.7, 1.5), p=0.2), B.RandomBrightness((0.7, 1.5), p=0.2),
            B.RandomGaussianBlur(p=0.1, kernel_size=(5, 5), sigma=(1.5, 1.5), silence_instantiation_warning=True),
            random_apply=4, generator=generator),
        B.KorniaImageDistortions(
            B.RandomGaussianNoise(std=4.0 / 255.0, p=0.25), B.RandomGaussianNoise(std=16.0 / 255.0, p=0.25 ** 2),
            B.RandomGaussianNoise(std=32.0 / 255.0, p=0.25 ** 3), B.RandomGaussianNoise(std=64.0 / 255.0, p=0.25 ** 4),
            B.OnlyClip(p=1.0), out_shift=-0.5 if whiten else 0.0, generator=generator),
    ]


class SyntheticPoseLoader:
    """Endless iterator of `list[Batch]` split by Tag in fixed proportions (the contract of the train
    loader returned by make_pose_estimation_loaders, reference :534-554)."""

    def __init__(self, batchsize: int, tags_and_weights: Sequence[tuple[Tag, float]], device="cuda", seed=1234, inputsize=129,
                 steps_per_epoch: int | None = None, image_augmentations=None, single_batch: bool = False):
        # single_batch: the contract of the TEST loader instead (reference :543-552: one Batch per iteration, not a list) - one Tag only
        if single_batch and len(tags_and_weights) != 1:
            raise ValueError("single_batch: one Tag")
        self._single = single_batch
        self._augs = image_augmentations  # containers from make_image_augmentations (images are un-whitened for them)
        total = sum(w for _, w in tags_and_weights)
        counts = [int(batchsize * w / total) for _, w in tags_and_weights]
        counts[0] += batchsize - sum(counts)
        self._plan = [(t, c) for (t, _), c in zip(tags_and_weights, counts) if c > 0]
        self._device, self._inputsize = device, inputsize
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(seed)
        self._steps = steps_per_epoch if steps_per_epoch is not None else (10 * 1024) // batchsize

    def __len__(self):
        return self._steps

    def __iter__(self) -> Iterator[list[Batch]]:
        for _ in range(self._steps):
            batches = [synthetic_subbatch(t, c, self._device, self._gen, self._inputsize) for t, c in self._plan]
            if self._augs:
                for b in batches:
                    img = b["image"] + 0.5  # synthetic crops are stored whitened
                    for aug in self._augs:
                        img = aug.apply(img, aug.sample_params(img.shape[0]))
                    b["image"] = img
            yield batches[0] if self._single else batches


# The pose datasets of the reference that carry what the pose-estimator step trains on (pipelines.py:120-300, 399-453): file (as an .npz
# shard converted by oracle/tools/h5_to_npz.py), task Tag, default sampling weight, and the frame range of the train split.
_POSE_SHARDS = {
    Id.REPO_300WLP: ("reproduction_300wlp-v12", Tag.POSE_WITH_LANDMARKS, 60_000.0, None),
    Id.REPO_300WLP_WO_EXTRA: ("reproduction_300wlp_simple", Tag.POSE_WITH_LANDMARKS, 60_000.0, None),
    Id._300WLP: ("300wlp", Tag.POSE_WITH_LANDMARKS_3D_AND_2D, 60_000.0, None),
    Id.WFLW_LP: ("wflw_augmented_v4", Tag.POSE_WITH_LANDMARKS, 40_000.0, None),
    Id.LAPA_MEGAFACE_LP: ("lapa-megaface-augmented-v2", Tag.POSE_WITH_LANDMARKS, 10_000.0, None),
    Id.REPLICANT_FACE: ("replicant-face-v4-wider-100k", Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS, 10_000.0, None),
    Id.BIWI: ("biwi-v3", Tag.ONLY_POSE, 1_000.0, None),
    Id.AFLW2k3d: ("aflw2k", Tag.POSE_WITH_LANDMARKS, 1_000.0, (400, None)),  # frames 400.. train, 0..399 test (:266-271)
}
_TEST_SHARD = ("aflw2k", Tag.POSE_WITH_LANDMARKS, (0, 400))  # the validation set of every run (:455-456)


# ---------------------------------------------------------------------------------------------
# validation sets of the evaluation script (reference :170-271, 557-636)
# ---------------------------------------------------------------------------------------------
# name -> (shard file, "filter").  The sets the reference builds from pose datasets with stored frames; the panoptic / replicant-face
# variants with a train/test split or a random subset are not listed.
_VALIDATION_SHARDS = {
    "aflw2k3d": ("aflw2k", "no_extreme_poses"),
    "aflw2k3d_closedeyes": ("aflw2k3d-closedeyes", "no_extreme_poses"),
    "aflw2k3d_grimaces": ("aflw2k", "grimaces"),
    "biwi": ("biwi-v3", None),
    "myself": ("myself", None),
    "myself_yaw": ("myself-yaw", None),
    "repro_300_wlp": ("reproduction_300wlp-v12", None),
    "wflw_lp": ("wflw_augmented_v4", None),
    "lapa_megaface_lp": ("lapa-megaface-augmented-v2", None),
    "replicantface": ("replicant-face-v4-eval-10k", None),
    "replicantface-stability": ("replicant-face-stability-test-wider", None),
}
# frames of AFLW2000-3D's first 400 (the test split) with strong facial expressions (reference :208-263)
_AFLW2K_GRIMACES = (39, 236, 0, 129, 164, 356, 359, 256, 136, 375, 226, 392, 119, 366, 293, 56, 305, 303, 397, 10, 11, 96, 173, 124, 115, 153, 337,
                    29, 121, 266, 387, 122, 8, 59, 108, 380, 187, 192, 353, 257, 162, 363, 331, 14, 163)


def indices_without_extreme_poses(quats, coords):
    """Frames whose AFLW-convention pitch, yaw and roll all stay below 99 degrees and whose size is not negative (reference :170-185)."""
    from scipy.spatial.transform import Rotation

    pyr = np.asarray([utils.inv_aflw_rotation_conversion(r) for r in Rotation.from_quat(np.asarray(quats))])
    ok = (np.abs(pyr) < np.pi * 99.0 / 180.0).all(axis=1) & (np.asarray(coords)[:, -1] >= 0.0)
    return np.nonzero(ok)[0]


class ValidationSamples:
    """Single labelled frames for `eval.Predictor.evaluate` (the reference's SampleBySampleLoader over make_validation_dataset): dicts with
    "image" (uint8 [H, W], unpadded), the labels as CPU tensors, "index" and - where the file knows it - "individual"."""

    def __init__(self, shard: dict, indices, put_roi):
        self._shard, self._indices, self._put_roi = shard, np.asarray(indices), put_roi

    def __len__(self):
        return len(self._indices)

    def __iter__(self):
        for i in self._indices:
            w, h = (int(v) for v in self._shard["image_size"][i])
            s = {k: torch.from_numpy(np.asarray(v[i])) for k, v in self._shard.items() if k not in ("image", "image_size")}
            s["image"] = torch.from_numpy(self._shard["image"][i, 0, :h, :w])
            s["index"] = torch.tensor(int(i), dtype=torch.int32)
            yield self._put_roi(s)


def make_validation_dataset(name, order=None, use_head_roi=True, datadir=None, headmodel=None) -> ValidationSamples:
    """Reference :557-606.  Every sample gets the half-pixel offset (datasets/shards.py) and `PutRoiFromLandmarks(extend_to_forehead=
    use_head_roi)`: use_head_roi=True - the reference's default, "(H_roi)" in its tables - needs the BFM head mesh (facemodel/bfm.py;
    FileNotFoundError without the blob), use_head_roi=False ("(F_roi)") takes the landmarks' extent."""
    import os

    from .datasets.shards import decode_pose_shard
    from .datatransformation.batch.misc import PutRoiFromLandmarks

    if name not in _VALIDATION_SHARDS:
        raise ValueError(f"unknown validation set {name!r} (known: {sorted(_VALIDATION_SHARDS)})")
    fname, rule = _VALIDATION_SHARDS[name]
    datadir = datadir or os.environ.get("DATADIR")
    if not datadir:
        raise RuntimeError("make_validation_dataset: set $DATADIR (or pass datadir=) to the directory of converted .npz shards")
    path = os.path.join(datadir, fname + ".npz")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: convert {fname}.h5 with oracle/tools/h5_to_npz.py --dataset")
    shard = decode_pose_shard(path)
    n = len(shard["image"])
    if rule == "no_extreme_poses":
        indices = indices_without_extreme_poses(shard["pose"], shard["coord"])
    elif rule == "grimaces":
        indices = np.asarray(_AFLW2K_GRIMACES)
    else:
        indices = np.arange(n)
    if order is not None:
        indices = indices[np.asarray(order)]
    return ValidationSamples(shard, indices, PutRoiFromLandmarks(extend_to_forehead=use_head_roi, headmodel=headmodel))


def make_validation_loader(name, order=None, use_head_roi=True, return_single_samples=True, datadir=None, headmodel=None):
    """Reference :608-636 with return_single_samples=True, the form the evaluation script uses (frames of a set differ in size)."""
    if not return_single_samples:
        raise NotImplementedError("batched validation loaders: use make_pose_estimation_loaders' test loader (HBM-resident, cropped on the GPU)")
    return make_validation_dataset(name, order, use_head_roi, datadir, headmodel)


def _slice_frames(frames, lo, hi):
    from .datasets.resident import ResidentFrames

    return ResidentFrames(frames.tag, {k: v[lo:hi] for k, v in frames.fields.items()})


def make_pose_estimation_loaders(inputsize, batchsize, datasets, dataset_weights=None, use_weights_as_sampling_frequency=True,
                                 enable_image_aug=True, rotation_aug_angle=30.0, roi_override="original", device="cuda", seed=1234,
                                 datadir=None, steps_per_epoch=None, headmodel=None, frames_on="auto", hbm_budget_bytes=None):
    """Signature of the reference (pipelines.py:359-369) plus `seed` (data-parallel replicas draw different streams), `datadir` and
    `steps_per_epoch`.  `datasets`:

      * "synthetic" (or a list of (Tag, weight) pairs): seeded synthetic loaders with the reference's contract;
      * a sequence of `Id`s: the reference's datasets from `datadir` (default $DATADIR) holding one `<name>.npz` shard per HDF5 file
        (oracle/tools/h5_to_npz.py converts them in the build container; datasets/shards.py decodes the JPEGs once).  The decoded frames
        live in HBM; per step the weighted concat draw of the reference (datasets/randomized.py), the random focus-ROI crop / warp and
        the intensity augmentation run on the GPU (datasets/resident.py).  The test loader is the deterministic crop of the first 400
        AFLW2000-3D frames, as in the reference.  Behind the crop every sample is mirrored with probability 1/2 and turned by +-90
        degrees with probability 0.5 % each (`horizontal_flip_and_rot_90(0.01)`, :373-377), composed into the crop's warp.
        `roi_override`: "original" (stored face boxes, crop enlargement 1.1) or "landmarks" (boxes = xy extent of pt3d_68 in front of and
        behind the crop, enlargement 1.2; :329-350) or "extent_to_forehead" (boxes = xy extent of the posed BFM head mesh in front of the
        crop only, enlargement 1.1; :351-356 - computed once per resident frame set, since they depend on the labels alone).  The last one
        needs the vertices of the full BFM head model, a blob neither the reference's repository nor this package carries
        (facemodel/bfm.py): FileNotFoundError without it.
        `frames_on`: "device" (decoded frames live in HBM), "host" (pinned host memory, gathered and copied per step on a side stream, one
        step ahead: datasets/resident.py) or "auto": in HBM while the decoded shards stay within `hbm_budget_bytes` (default: half of
        the device's memory), the largest datasets on the host beyond that.
    """
    if datasets == "synthetic":
        datasets = [(Tag.POSE_WITH_LANDMARKS, 11.0), (Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS, 1.0)]
    if isinstance(datasets, (list, tuple)) and datasets and isinstance(datasets[0], tuple) and isinstance(datasets[0][0], Tag):
        augs = make_image_augmentations(torch.Generator().manual_seed(99 + seed)) if enable_image_aug and str(device).startswith("cuda") else None
        train = SyntheticPoseLoader(batchsize, datasets, device=device, seed=seed, inputsize=inputsize, image_augmentations=augs)
        test = SyntheticPoseLoader(batchsize, [(Tag.POSE_WITH_LANDMARKS, 1.0)], device=device, seed=4321, inputsize=inputsize,
                                   steps_per_epoch=max(1, 400 // batchsize), single_batch=True)
        return train, test, len(train) * batchsize
    if not (isinstance(datasets, (list, tuple)) and datasets and all(isinstance(d, Id) for d in datasets)):
        raise ValueError('datasets: "synthetic", a list of (Tag, weight) pairs, or a sequence of pipelines.Id')
    if roi_override not in ("original", "landmarks", "extent_to_forehead"):
        raise ValueError(f"roi_override: got {roi_override!r}")  # (the reference asserts, :330)
    head_roi = None
    if roi_override == "extent_to_forehead":
        from .datatransformation.batch.misc import PutRoiFromLandmarks

        head_roi = PutRoiFromLandmarks(extend_to_forehead=True, headmodel=headmodel)  # FileNotFoundError when the BFM blob is absent
    extension_factor = {"original": 1.1, "extent_to_forehead": 1.1, "landmarks": 1.2}[roi_override]  # :333
    import os

    from .datasets.resident import ResidentEvalLoader, ResidentLoader
    from .datasets.shards import load_resident_frames
    from .datatransformation.gpu import GpuFocusRoiAugment

    datadir = datadir or os.environ.get("DATADIR")
    if not datadir:
        raise RuntimeError("make_pose_estimation_loaders: set $DATADIR (or pass datadir=) to the directory of converted .npz shards")
    unsupported = [d for d in datasets if d not in _POSE_SHARDS]
    if unsupported:
        raise NotImplementedError(f"datasets {unsupported}: landmark-only / face-detection / segmentation sets are outside the pose-estimator "
                                  f"path this package implements (supported: {sorted(d.name for d in _POSE_SHARDS)})")
    if len([d for d in datasets if d in (Id._300WLP, Id.REPO_300WLP, Id.REPO_300WLP_WO_EXTRA)]) > 1:
        raise ValueError("at most one 300W-LP variant (reference :435-438)")
    if frames_on not in ("auto", "device", "host"):
        raise ValueError(f"frames_on: got {frames_on!r}")
    cache: dict = {}
    budget = hbm_budget_bytes
    if budget is None and frames_on == "auto":
        budget = torch.cuda.get_device_properties(device).total_memory // 2 if str(device).startswith("cuda") else 0
    on_device_bytes = [0]

    def shard(name, tag):
        path = os.path.join(datadir, name + ".npz")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found: convert {name}.h5 with `/opt/conda/bin/python3.9 oracle/tools/h5_to_npz.py --dataset "
                                    f"{name}.h5 {path}` (h5py lives in the build container's conda interpreter only)")
        if path not in cache:
            frames = load_resident_frames(path, tag, "cpu")
            if head_roi is not None and "pt3d_68" in frames.fields and str(device).startswith("cuda"):
                # the forehead box is the extent of the posed head MESH (tens of thousands of vertices per frame): computed on the GPU from
                # the labels alone, BEFORE the frames are placed - on host-placed sets it would otherwise run on the CPU for every frame
                labels = {k: v.to(device) for k, v in frames.fields.items() if k != "image"}
                head_roi(labels)
                frames.fields["roi"] = labels["roi"].cpu()
            elif head_roi is not None:
                head_roi(frames.fields)  # frames with landmarks get the forehead box; the others keep their stored one
            if frames_on == "device" or (frames_on == "auto" and on_device_bytes[0] + frames.nbytes() <= budget):
                on_device_bytes[0] += frames.nbytes()
                cache[path] = frames.to(device)
            else:  # beyond the HBM budget (or asked for): pinned host memory, streamed per step
                cache[path] = frames.to_host()
        return cache[path]

    dataset_weights = dataset_weights or {}
    train_sets, weights = [], []
    for d in datasets:
        name, tag, default_w, rng = _POSE_SHARDS[d]
        frames = shard(name, tag)
        if rng is not None:
            frames = _slice_frames(frames, rng[0], rng[1])
        train_sets.append(frames)
        weights.append(float(dataset_weights.get(d, default_w)))
    total = sum(len(t) for t in train_sets)
    if use_weights_as_sampling_frequency:
        freqs = [w / sum(weights) for w in weights]
    else:  # the weights scale the losses instead (`dataset_weight` field, reference :475-485); every dataset is drawn equally often
        wmax = max(weights)
        for t, w in zip(train_sets, weights):
            # beside the set's other fields: on the device, or - host-placed sets - in pinned host memory (the host gather reads numpy views)
            col = torch.full((len(t),), w / wmax, dtype=torch.float32, device=t.fields["image"].device)
            t.fields["dataset_weight"] = col.pin_memory() if (col.device.type == "cpu" and torch.cuda.is_available()) else col
        freqs = [1.0 / len(weights)] * len(weights)
    augs = make_image_augmentations(torch.Generator().manual_seed(99 + seed)) if enable_image_aug else None
    crop = GpuFocusRoiAugment(new_size=inputsize, rotation_aug_angle=rotation_aug_angle, extension_factor=extension_factor, whiten=not augs,
                              flip_rot_p=0.01, roi_from_landmarks=roi_override == "landmarks")
    steps = steps_per_epoch if steps_per_epoch is not None else (10 * 1024) // batchsize  # Trainer(limit_train_batches=...), train_poseestimator.py:447
    train = ResidentLoader(train_sets, freqs, batchsize, steps, seed=seed, crop=crop, image_augmentations=augs, device=device)
    tname, ttag, (lo, hi) = _TEST_SHARD
    test = ResidentEvalLoader([_slice_frames(shard(tname, ttag), lo, hi)], batchsize * 2, new_size=inputsize, extension_factor=extension_factor,
                              roi_from_landmarks=roi_override == "landmarks", device=device)
    return train, test, total
