"""Rotation conventions and small helpers (reference: trackertraincode/utils.py)."""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

rad2deg = 180.0 / np.pi
deg2rad = np.pi / 180.0
_P = np.asarray([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]])


def convert_to_rot(net_output) -> Rotation:
    """Quaternions (i, j, k, w) -> scipy Rotation (reference :37-38)."""
    return Rotation.from_quat(net_output)


def aflw_rotation_conversion(pitch, yaw, roll) -> Rotation:
    """AFLW / 300W-LP Euler angles -> Rotation (reference :41-51)."""
    rot = Rotation.from_euler("XYZ", np.asarray([pitch, -yaw, roll]).T)
    return Rotation.from_matrix(_P @ rot.as_matrix() @ _P.T)


def inv_aflw_rotation_conversion(rot: Rotation):
    """Rotation -> (pitch, yaw, roll) in the AFLW2000-3D convention (reference :53-64)."""
    euler = Rotation.from_matrix(_P @ rot.as_matrix() @ _P.T).as_euler("XYZ")
    return euler * np.asarray([1.0, -1.0, 1.0])


def iter_batched(iterable, batchsize):
    """Lists of up to `batchsize` items (reference :79-89)."""
    if isinstance(iterable, np.ndarray):
        for i in range(0, iterable.shape[0], batchsize):
            yield iterable[i:i + batchsize, ...]
        return
    it = iter(iterable)
    while True:
        ret = [x for _, x in zip(range(batchsize), it)]
        if not ret:
            break
        yield ret


def cycle(iterable):
    """Endless iteration that re-creates the iterator instead of caching its items (reference :92-100)."""
    while True:
        empty = True
        for x in iterable:
            empty = False
            yield x
        if empty:
            return


def num_workers() -> int:
    """Worker processes for CPU-side loading (reference :107-114: NUM_WORKERS env or the core count)."""
    import os

    return int(os.environ.get("NUM_WORKERS", os.cpu_count() or 1))
