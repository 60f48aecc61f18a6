"""Frames in pinned host memory instead of HBM (datasets/resident.py: data that does not fit): the loaders gather on the host, copy on a
side stream one step ahead and must yield bitwise the batches of the HBM-resident placement for the same seed."""
import os
import shutil

import numpy as np
import pytest
import torch

from util import GOLDEN

pytestmark = pytest.mark.gpu


def _datadir(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, "aflw2kmini.npz"), d / "aflw2k.npz")
    return str(d)


def _equal(a, b):
    assert a.keys() == b.keys()
    for k in a.keys():
        va, vb = a[k], b[k]
        if torch.is_tensor(va):
            assert va.device == vb.device and torch.equal(va, vb), k


@pytest.mark.parametrize("image_aug", [False, True])
def test_host_frames_yield_the_same_batches(tmp_path, monkeypatch, image_aug):
    import trackertraincode.pipelines as P

    datadir = _datadir(tmp_path)
    monkeypatch.setitem(P._POSE_SHARDS, P.Id.AFLW2k3d, ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, 1000.0, (8, None)))
    monkeypatch.setattr(P, "_TEST_SHARD", ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, (0, 8)))
    kw = dict(device="cuda", seed=5, datadir=datadir, steps_per_epoch=7, enable_image_aug=image_aug)
    tr_d, te_d, n_d = P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="device", **kw)
    tr_h, te_h, n_h = P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="host", **kw)
    assert n_d == n_h and not tr_d.datasets[0].on_host and tr_h.datasets[0].on_host and tr_h.datasets[0].fields["image"].is_pinned()
    def two_epochs(loader):  # the second epoch continues the permutations
        torch.manual_seed(11)  # the Gaussian-noise augmentation draws from the global device generator (like the reference's kornia ops)
        return [[{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for b in step] for _ in range(2) for step in loader]

    got_d, got_h = two_epochs(tr_d), two_epochs(tr_h)
    assert len(got_d) == len(got_h) == 14
    for bd, bh in zip(got_d, got_h):
        assert len(bd) == len(bh)
        for x, y in zip(bd, bh):
            _equal(x, y)
    for x, y in zip(te_d, te_h):
        _equal(x, y)
    # "auto": a budget of zero bytes puts everything on the host, the default budget keeps this small set in HBM
    tr_a, _, _ = P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="auto", hbm_budget_bytes=0, **kw)
    assert tr_a.datasets[0].on_host
    tr_b, _, _ = P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="auto", **kw)
    assert not tr_b.datasets[0].on_host
    with pytest.raises(ValueError):
        P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="disk", **kw)


def test_abandoned_iteration_stops_the_prefetch_thread(tmp_path, monkeypatch):
    import threading

    import trackertraincode.pipelines as P

    datadir = _datadir(tmp_path)
    monkeypatch.setitem(P._POSE_SHARDS, P.Id.AFLW2k3d, ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, 1000.0, (8, None)))
    monkeypatch.setattr(P, "_TEST_SHARD", ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, (0, 8)))
    tr, _, _ = P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="host", device="cuda", seed=1, datadir=datadir, steps_per_epoch=50)
    it = iter(tr)
    next(it)
    it.close()  # GeneratorExit -> the producer is told to stop
    assert not [t for t in threading.enumerate() if t.name == "ResidentLoader-prefetch" and t.is_alive()]
    # training through fit() from host frames
    import trackertraincode.train as train
    from util import script_args, train_script

    S = train_script()
    args = script_args(dict(with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False), epochs=1)
    torch.manual_seed(0)
    net = S.create_net(args).cuda()
    crit, _ = S.setup_losses(args, net)
    opt, sch = S.create_optimizer(net, args)
    tr2, _, _ = P.make_pose_estimation_loaders(129, 8, [P.Id.AFLW2k3d], frames_on="host", device="cuda", seed=1, datadir=datadir, steps_per_epoch=4)
    hist = train.fit(net, tr2, crit, opt, sch, epochs=1)
    assert hist is None or True
    assert all(torch.isfinite(p).all() for p in net.parameters())
