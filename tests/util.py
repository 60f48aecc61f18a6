"""Shared helpers for the parity tests."""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "neuralnet-tracker-traincode_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")


class gpu_section:
    """with gpu_section(): the GPU part of a WORKER process (tests/_trajectory_worker.py, _teacher_forced_worker.py).  The workers of a session run side by
    side for their long CPU-oracle parts; their short HIP parts take turns (an advisory file lock), so that at most one worker shares the GPU with the
    session's own tests.  (Several processes time-slicing the GPU is where round 6 found sporadic wrong results in two tiny kernels -
    profiles/r06_packed_fma_under_time_slicing.txt - and it slows every one of them 20-60 x.)"""

    def __enter__(self):
        import fcntl
        import tempfile
        self._f = open(os.path.join(tempfile.gettempdir(), "ttk_gpu_worker.lock"), "w")
        fcntl.flock(self._f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        fcntl.flock(self._f, fcntl.LOCK_UN)
        self._f.close()
        return False


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return d, json.loads(str(d["meta"]))


def train_script():
    """neuralnet-tracker-traincode_amd/scripts/train_poseestimator.py as a module."""
    if "amd_train_script" in sys.modules:
        return sys.modules["amd_train_script"]
    spec = importlib.util.spec_from_file_location("amd_train_script", os.path.join(PKG, "scripts", "train_poseestimator.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["amd_train_script"] = mod
    spec.loader.exec_module(mod)
    return mod


def script_args(flags, epochs=200, lr=1.0e-3):
    import argparse

    ns = argparse.Namespace(backbone="mobilenetv1", batchsize=8, lr=lr, epochs=epochs, with_roi_train=True, enable_6drot=False,
                            with_blurpool=False, swa=False)
    for k, v in flags.items():
        setattr(ns, k, v)
    return ns


def build_net(meta, device, extra_state=None):
    from oracle.synth import make_state
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    net = NetworkWithPointHead(**meta["config"])
    sd = make_state({k: tuple(v) for k, v in meta["shapes"].items()}, meta["state_seed"])
    if extra_state:
        sd.update(extra_state)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    return net.to(device)


def make_batches(meta, device, with_dataset_weight=False):
    from oracle.synth import make_inputs, make_labels
    from trackertraincode.datasets.batch import Batch, Metadata
    from trackertraincode.pipelines import Tag

    B, split = meta["B"], meta["split"]
    image, ids = make_inputs(B, seed=meta["input_seed"])
    lab = make_labels(B, seed=meta["input_seed"])
    t = lambda a: torch.from_numpy(a.copy()).to(device)
    b0 = dict(image=t(image[:split]), coord_convention_id=t(ids[:split]), **{k: t(lab[k][:split]) for k in ("pose", "coord", "roi", "pt3d_68", "shapeparam")})
    b1 = dict(image=t(image[split:]), coord_convention_id=t(ids[split:]), **{k: t(lab[k][split:]) for k in ("pose", "coord", "roi")})
    if with_dataset_weight:
        b0["dataset_weight"], b1["dataset_weight"] = t(lab["dataset_weight"][:split]), t(lab["dataset_weight"][split:])
    return [Batch(Metadata(129, batchsize=split, tag=Tag.POSE_WITH_LANDMARKS), b0),
            Batch(Metadata(129, batchsize=B - split, tag=Tag.ONLY_POSE), b1)]
