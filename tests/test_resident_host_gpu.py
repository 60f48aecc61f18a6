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


def test_mixed_placement_three_datasets_two_sizes():
    """Three datasets (two Tags, two frame sizes), one of them on the host, the others in HBM: the same batches as with everything in HBM."""
    from trackertraincode.datasets.resident import ResidentFrames, ResidentLoader
    from trackertraincode.datatransformation.gpu import GpuFocusRoiAugment
    from trackertraincode.pipelines import Tag

    def frames(tag, n, size, seed, pts):
        g = torch.Generator().manual_seed(seed)
        f = {"image": torch.randint(0, 255, (n, 1, size, size), dtype=torch.uint8, generator=g),
             "roi": torch.tensor([[0.25 * size, 0.25 * size, 0.75 * size, 0.75 * size]]).repeat(n, 1) + torch.randn(n, 4, generator=g) * 2,
             "coord": torch.cat([torch.full((n, 2), 0.5 * size), torch.full((n, 1), 0.2 * size)], -1),
             "pose": torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1),
             "coord_convention_id": torch.zeros(n, dtype=torch.int32)}
        if pts:
            f["pt3d_68"] = torch.rand(n, 68, 3, generator=g) * size
            f["shapeparam"] = torch.randn(n, 50, generator=g)
        return ResidentFrames(tag, f)

    host_sets = [frames(Tag.POSE_WITH_LANDMARKS, 90, 96, 1, True), frames(Tag.POSE_WITH_LANDMARKS, 70, 128, 2, True), frames(Tag.ONLY_POSE, 50, 80, 3, False)]

    def run(placement):
        sets = [s.to("cuda") if on == "d" else s.to_host() for s, on in zip(host_sets, placement)]
        crop = GpuFocusRoiAugment(new_size=129, rotation_aug_angle=25.0, extension_factor=1.1, whiten=True, flip_rot_p=0.01)
        loader = ResidentLoader(sets, [0.5, 0.3, 0.2], 48, 9, seed=4, crop=crop)
        return [[{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for b in step] for step in loader]

    ref, mixed, allhost = run("ddd"), run("dhd"), run("hhh")
    for other in (mixed, allhost):
        assert len(other) == len(ref) == 9
        for sa, sb in zip(ref, other):
            assert len(sa) == len(sb) == 2  # two Tags; the two landmark sets differ in frame size: cropped per set, collated
            for x, y in zip(sa, sb):
                _equal(x, y)


def test_host_frames_with_loss_weights_and_a_captured_step(tmp_path, monkeypatch):
    """Round-4 advisor findings: (1) `use_weights_as_sampling_frequency=False` adds a `dataset_weight` column to every train set - on a
    host-placed set it must sit beside the host fields (the host gather reads numpy views), and the loaders must use the requested device;
    (2) the prefetch thread of host-placed sets allocates and copies while `fit(graphed=True)` captures the step as a hipGraph - both take
    _hip.CAPTURE_LOCK."""
    import trackertraincode.pipelines as P
    import trackertraincode.train as train
    from util import script_args, train_script

    datadir = _datadir(tmp_path)
    monkeypatch.setitem(P._POSE_SHARDS, P.Id.AFLW2k3d, ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, 1000.0, (8, None)))
    monkeypatch.setattr(P, "_TEST_SHARD", ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, (0, 8)))
    kw = dict(device="cuda", seed=5, datadir=datadir, steps_per_epoch=5, enable_image_aug=False, use_weights_as_sampling_frequency=False)
    tr_d, _, _ = P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="device", **kw)
    tr_h, _, _ = P.make_pose_estimation_loaders(129, 6, [P.Id.AFLW2k3d], frames_on="host", **kw)
    assert tr_h.datasets[0].on_host and tr_h.datasets[0].fields["dataset_weight"].device.type == "cpu" and tr_h._device.type == "cuda"
    for step_d, step_h in zip(tr_d, tr_h):
        for x, y in zip(step_d, step_h):
            assert "dataset_weight" in x.keys() and x["dataset_weight"].is_cuda
            _equal(x, y)
    # a captured training step fed from host frames (the producer thread runs beside the capture)
    S = train_script()
    args = script_args(dict(with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False), epochs=2)
    torch.manual_seed(0)
    net = S.create_net(args).cuda()
    crit, _ = S.setup_losses(args, net)
    opt, sch = S.create_optimizer(net, args)
    tr2, _, _ = P.make_pose_estimation_loaders(129, 8, [P.Id.AFLW2k3d], frames_on="host", device="cuda", seed=1, datadir=datadir, steps_per_epoch=6)
    train.fit(net, tr2, crit, opt, sch, epochs=2, graphed=True)
    assert all(torch.isfinite(p).all() for p in net.parameters())
