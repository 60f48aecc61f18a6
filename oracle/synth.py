"""Deterministic synthetic weights / inputs / labels shared by the golden generator, the tests and
bench.py's cpu_baseline leg.

TEST INFRASTRUCTURE (part of oracle/): only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package.

Everything is drawn from numpy's PCG64 (`np.random.default_rng`), whose stream is specified and
platform-independent, keyed by (seed, crc32(name)) so that values do not depend on generation order.
Distributions follow SURVEY.md §8(d) ("Concrete synthetic inputs").
"""
from __future__ import annotations

import math
import zlib

import numpy as np

F32 = np.float32


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


# ---------------------------------------------------------------------------------------------
# 3DMM keypoint buffers.  The real blob (facemodel/bfm_noneck_v3.pkl) is missing from the reference
# checkout (.MISSING_LARGE_BLOBS:2), so `keypts`/`keyeigvecs` are SYNTHETIC state-dict buffers.
# Shapes follow modelcomponents.py:65-69.
# ---------------------------------------------------------------------------------------------
_BFM_V = 256


def synthetic_bfm_arrays(shape_dim: int = 40, exp_dim: int = 10):
    rng = np.random.default_rng(20240711)
    vertices = (rng.standard_normal((_BFM_V, 3)) * 0.5).astype(F32)
    bases = (rng.standard_normal((shape_dim + exp_dim, _BFM_V, 3)) * 0.05).astype(F32)
    keypoints = rng.permutation(_BFM_V)[:68].astype(np.int64)
    return vertices, bases, keypoints


def synthetic_keypoint_buffers():
    """(keypts[68,3], keyeigvecs[50,68,3])"""
    vertices, bases, kp = synthetic_bfm_arrays()
    return np.ascontiguousarray(vertices[kp]), np.ascontiguousarray(bases[:, kp, :])


# ---------------------------------------------------------------------------------------------
# Weights
# ---------------------------------------------------------------------------------------------
_BIAS_BASE = {
    # init overrides of the reference heads (models.py:132,159,182,206; negloglikelihood.py:28,231)
    "boxnet.linear.bias": [0.0, 0.0, 0.5, 0.5],
    "posnet.linear_size.bias": [0.5],
    "quatnet.linear.bias": [0.0, 0.0, 0.0, math.log(0.1)],
}


def make_state(shapes: dict[str, tuple], seed: int = 0) -> dict[str, np.ndarray]:
    """Fill a state dict (name -> shape) with deterministic, well-conditioned values."""
    out: dict[str, np.ndarray] = {}
    kp, ke = synthetic_keypoint_buffers()
    for name, shape in shapes.items():
        shape = tuple(shape)
        r = _rng(seed, name)
        leaf = name.rsplit(".", 1)[-1]
        is_bn = (".bn" in name or name.startswith("bn") or ".downsample.1" in name
                 or name.startswith("layers.1.") or ".layers.1." in name)  # layers.1 = the ResNet stem's BatchNorm
        if leaf == "num_batches_tracked":
            v = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            v = (r.standard_normal(shape) * 0.05).astype(F32)
        elif leaf == "running_var":
            v = r.uniform(0.5, 1.5, shape).astype(F32)
        elif leaf == "keypts":
            v = kp
        elif leaf == "keyeigvecs":
            v = ke
        elif leaf == "min_diag":
            v = np.array([1e-6] * 3 + [0.0] * 3, dtype=F32)
        elif leaf == "kernel":  # BlurPool2D buffer (modelcomponents.py:197)
            row = np.array([1.0, 2.0, 1.0])
            v = (np.outer(row, row) / 16.0).astype(F32).reshape(shape)
        elif len(shape) == 4:  # conv weight, mobilenet_v1.py:155-158 init law
            n = shape[2] * shape[3] * shape[0]
            v = (r.standard_normal(shape) * math.sqrt(2.0 / n)).astype(F32)
        elif is_bn and leaf == "weight":
            v = r.uniform(0.5, 1.5, shape).astype(F32)
        elif is_bn and leaf == "bias":
            v = (r.standard_normal(shape) * 0.1).astype(F32)
        elif len(shape) == 2 and leaf == "weight":  # linear
            v = (r.standard_normal(shape) * 0.02).astype(F32)
        elif leaf == "p":  # LocalToGlobalCoordinateOffset.p (8,4)
            v = (r.standard_normal(shape) * 0.1).astype(F32)
        elif leaf in ("bias", "hidden_scale"):
            base = np.asarray(_BIAS_BASE.get(name, 0.0), dtype=F32)
            if name == "quatnet.linear.bias" and shape == (6,):
                # 6D head (models.py:159 biases towards identity, x=(1,0,0), y=(0,1,0)); unit scale here so that the
                # synthetic x and y stay well away from collinear
                base = np.asarray([1.0, 0.0, 0.0, 0.0, 1.0, 0.0], dtype=F32)
            v = (np.broadcast_to(base, shape) + r.standard_normal(shape) * 0.05).astype(F32)
        else:
            raise KeyError(f"no synthetic rule for state entry {name} {shape}")
        assert tuple(v.shape) == shape, (name, v.shape, shape)
        out[name] = v
    return out


def make_grads(shapes: dict[str, tuple], seed: int, scale: float) -> dict[str, np.ndarray]:
    """Synthetic gradients N(0, scale^2) per parameter, for the optimiser known-answer tests."""
    return {
        k: (_rng(seed, "grad:" + k).standard_normal(tuple(v)) * scale).astype(F32) for k, v in shapes.items()
    }


# ---------------------------------------------------------------------------------------------
# Inputs and labels
# ---------------------------------------------------------------------------------------------
def make_inputs(batch: int, seed: int = 1234, resolution: int = 129, structured: bool = True):
    """image[B,1,R,R] f32 in [-0.5,0.5], coord_convention_id[B] int32 in 0..7.

    structured=False: `rand - 0.5` (the bench workload).
    structured=True : per-sample mix of plane waves + noise, so that the pooled features differ
    between samples (uniform noise averages out and would make parity checks insensitive)."""
    r = _rng(seed, "image")
    image = (r.random((batch, 1, resolution, resolution), dtype=F32) - F32(0.5)).astype(F32)
    if structured:
        yy, xx = np.meshgrid(np.arange(resolution), np.arange(resolution), indexing="ij")
        waves = np.zeros((batch, resolution, resolution), dtype=np.float64)
        for _ in range(3):
            fx = r.uniform(-0.25, 0.25, (batch, 1, 1))
            fy = r.uniform(-0.25, 0.25, (batch, 1, 1))
            ph = r.uniform(0, 2 * np.pi, (batch, 1, 1))
            am = r.uniform(0.05, 0.25, (batch, 1, 1))
            waves += am * np.sin(fx * xx[None] + fy * yy[None] + ph)
        image = np.clip(0.3 * image + waves[:, None].astype(F32), -0.5, 0.5).astype(F32)
    ids = _rng(seed, "ids").integers(0, 8, size=(batch,)).astype(np.int32)
    return image, ids


def make_labels(batch: int, seed: int = 1234):
    r = _rng(seed, "labels")
    pose = r.standard_normal((batch, 4)).astype(F32)
    pose /= np.linalg.norm(pose, axis=-1, keepdims=True)
    coord = np.stack(
        [r.uniform(-0.3, 0.3, batch), r.uniform(-0.3, 0.3, batch), r.uniform(0.8, 1.6, batch)], axis=-1
    ).astype(F32)
    roi = (np.array([-0.85, -0.85, 0.85, 0.85]) + r.uniform(-0.1, 0.1, (batch, 4))).astype(F32)
    pt3d_68 = (r.standard_normal((batch, 68, 3)) * 0.5).astype(F32)
    shapeparam = (r.standard_normal((batch, 50)) * 0.5).astype(F32)
    dataset_weight = r.uniform(0.5, 2.0, (batch,)).astype(F32)
    return {
        "pose": pose,
        "coord": coord,
        "roi": roi,
        "pt3d_68": pt3d_68,
        "shapeparam": shapeparam,
        "dataset_weight": dataset_weight,
    }


# ---------------------------------------------------------------------------------------------
# Compact digests so that multi-megabyte tensors can be pinned by small fixtures.
# ---------------------------------------------------------------------------------------------
def digest(a, nsample: int = 96) -> np.ndarray:
    """[l2-norm, sum, abs-sum, n, sample_0 .. sample_{k-1}] in float64; samples at fixed strided
    flat indices.  Small tensors (<= nsample elements) are stored whole."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    n = a.size
    idx = np.unique(np.linspace(0, max(n - 1, 0), num=min(nsample, n)).astype(np.int64))
    head = np.array([np.sqrt((a * a).sum()), a.sum(), np.abs(a).sum(), float(n)])
    return np.concatenate([head, a[idx]])


def digest_close(d_expected: np.ndarray, a, rtol: float, atol: float, rtol_samples: float | None = None):
    """Compare a tensor with a stored digest. Returns (ok, message).  `rtol_samples` (default: rtol) is the
    tolerance of the strided samples relative to max(rms, largest sample); `rtol` that of the l2 norm."""
    rtol_s = rtol if rtol_samples is None else rtol_samples
    d = digest(a, nsample=max(len(d_expected) - 4, 1))
    if d.shape != d_expected.shape:
        return False, f"digest shape {d.shape} vs {d_expected.shape}"
    scale = max(d_expected[0] / math.sqrt(max(d_expected[3], 1.0)), 1e-30)  # rms of the tensor
    err_s = np.abs(d[4:] - d_expected[4:]).max() if len(d) > 4 else 0.0
    ok_s = err_s <= atol + rtol_s * max(scale, np.abs(d_expected[4:]).max() if len(d) > 4 else 0.0)
    err_n = abs(d[0] - d_expected[0])
    ok_n = err_n <= atol * math.sqrt(d_expected[3]) + rtol * d_expected[0]
    return bool(ok_s and ok_n), f"sample err {err_s:.3e} (rms {scale:.3e}), norm err {err_n:.3e} of {d_expected[0]:.3e}"


# ---------------------------------------------------------------------------------------------
# A seeded synthetic blob in the format of the reference's missing bfm_noneck_v3.pkl / tri.pkl (3DDFA_V2): test input for
# facemodel/bfm.py.  V must exceed the hard-coded eye-contour vertex indices (14 327).
# ---------------------------------------------------------------------------------------------
def write_synthetic_bfm_blob(folder: str, V: int = 14400, n_shp: int = 42, n_exp: int = 11, seed: int = 515) -> None:
    import os
    import pickle

    rng = np.random.default_rng(seed)
    u = (rng.standard_normal((3 * V, 1)) * 4.0e4 + np.tile([0.0, 2.6e4, 9.0e4], V)[:, None]).astype(np.float64)  # the loader casts to f32
    w_shp = (rng.standard_normal((3 * V, n_shp)) * 3.0e-4).astype(np.float64)
    w_exp = (rng.standard_normal((3 * V, n_exp)) * 40.0).astype(np.float64)
    kp = rng.permutation(V)[:68].astype(np.int64)
    keypoints = (3 * kp[:, None] + np.arange(3)[None, :]).reshape(-1)
    tri = rng.integers(0, V, (3, 64)).astype(np.int64)
    with open(os.path.join(folder, "bfm_noneck_v3.pkl"), "wb") as f:
        pickle.dump({"u": u, "w_shp": w_shp, "w_exp": w_exp, "keypoints": keypoints}, f)
    with open(os.path.join(folder, "tri.pkl"), "wb") as f:
        pickle.dump(tri, f)
