"""Import shims for running the REFERENCE (/root/reference) inside the build container.

TEST INFRASTRUCTURE, build-container only. Never imported by the product, by `-m gpu` tests, by
`smoke()` or by `bench.py`; /root/reference does not exist on the GPU box.

The reference's third-party imports h5py, strenum, kornia, pytorch_lightning, torchvision, cv2,
torchmetrics are not installed here and cannot be installed (no network).  None of them takes
part in the arithmetic of the hot path (model forward/backward + losses + Adam): they provide a
StrEnum base, a Lightning base class, names of augmentations, and two kornia helpers used only when
`use_blurpool=True`.  `install()` registers name-only stand-in *modules* in `sys.modules` so that
the reference's own source files import and run unmodified on torch-CPU.  This follows SURVEY.md
Appendix E.  Nothing here stands in for a header, library or tool of a *compiled* reference build;
the arithmetic executed through this import is the reference's own Python + torch CPU kernels.

The missing `bfm_noneck_v3.pkl` blob (.MISSING_LARGE_BLOBS) makes `BFMModel()` unconstructible;
`install(synthetic_bfm=True)` patches `trackertraincode.facemodel.bfm.BFMModel` with a seeded
synthetic 68-keypoint basis (documented as synthetic in every fixture that uses it).
"""
from __future__ import annotations

import enum
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


class _Anything:
    """Placeholder class handed out for any attribute of a stand-in module."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("stand-in object called: this third-party feature is not available")

    def __class_getitem__(cls, item):
        return cls


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (_Anything,), {})
        setattr(self, name, cls)
        return cls


def _stub(name: str, **attrs) -> types.ModuleType:
    m = _StubModule(name)
    m.__path__ = []  # behave like a package so that sub-imports resolve through sys.modules
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


def _pascal_kernel_2d(kernel_size, norm=True, *, device=None, dtype=None):
    # Binomial (pascal-triangle) outer product; what kornia.filters.kernels.get_pascal_kernel_2d
    # returns for an int kernel size.  Only reached with use_blurpool=True (off in BASELINE configs).
    import math
    import torch

    ks = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
    row = torch.tensor([math.comb(ks - 1, i) for i in range(ks)], dtype=torch.float32)
    k = row[:, None] * row[None, :]
    if norm:
        k = k / k.sum()
    return k


def _blur_pool_by_kernel2d(input, kernel, stride):
    import torch.nn.functional as F

    ks = kernel.shape[-1]
    return F.conv2d(input, kernel, padding=(ks - 1) // 2, stride=stride, groups=input.shape[1])


class SyntheticBFM:
    """Stand-in for facemodel/bfm.py:23-97 exposing only what DeformableHeadKeypoints reads
    (modelcomponents.py:65-69): `scaled_vertices[V,3]`, `scaled_bases[50,V,3]`, `keypoints[68]`.
    Values are SYNTHETIC (seeded, oracle/synth.py), the real blob is missing from the reference."""

    def __init__(self, shape_dim=40, exp_dim=10):
        from oracle.synth import synthetic_bfm_arrays

        self.scaled_vertices, self.scaled_bases, self.keypoints = synthetic_bfm_arrays(
            shape_dim, exp_dim
        )


def install(synthetic_bfm: bool = True, gmm_npz: str | None = None):
    import torch
    import torch.nn as nn

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    repo_root = __file__.rsplit("/oracle/", 1)[0]
    if repo_root not in sys.path:
        sys.path.insert(1, repo_root)

    # h5py: names only.  The one data file on the hot path (shapeparams_gmm.h5) is replaced by a
    # File stand-in that serves the arrays extracted by oracle/tools/h5_to_npz.py.
    class _Attrs(dict):
        pass

    class _File:
        def __init__(self, filename, mode="r"):
            if not str(filename).endswith("shapeparams_gmm.h5") or gmm_npz is None:
                raise RuntimeError(f"h5py stand-in cannot open {filename}")
            self._d = dict(np.load(gmm_npz))
            self.attrs = _Attrs(covariance_type="diag")

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __getitem__(self, k):
            return self._d[k]

    _stub("h5py", File=_File)

    class StrEnum(str, enum.Enum):
        pass

    _stub("strenum", StrEnum=StrEnum)

    _stub("kornia")
    _stub("kornia.filters")
    _stub("kornia.filters.kernels", get_pascal_kernel_2d=_pascal_kernel_2d)
    _stub("kornia.filters.blur_pool", _blur_pool_by_kernel2d=_blur_pool_by_kernel2d)
    _stub("kornia.augmentation")

    class _Callback:
        pass

    _stub("pytorch_lightning", LightningModule=nn.Module, Callback=_Callback)
    _stub("pytorch_lightning.callbacks", Callback=_Callback)

    _stub("torchvision")
    _stub("torchvision.models")
    _stub("torchvision.models.resnet")
    _stub("torchvision.models.efficientnet")
    _stub("torchvision.models.mnasnet")
    _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional")
    _stub("cv2")
    _stub("torchmetrics")
    _stub("mkl")

    if synthetic_bfm:
        import trackertraincode.facemodel.bfm as bfm

        bfm.BFMModel = SyntheticBFM
    return torch
