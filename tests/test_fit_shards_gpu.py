"""End to end from converted shards to checkpoints on the MI355X (SURVEY.md §8 f2 + the harness around the path):

  aflw2kmini.npz (the reference's bundled aflw2kmini.h5 through oracle/tools/h5_to_npz.py: 16 JPEG frames + labels)
    -> datasets/shards.py (PIL decode, padding, name mapping, half-pixel offset) -> frames resident in HBM
    -> make_pose_estimation_loaders(datasets=[Id.AFLW2k3d], datadir=...) : weighted draw, random focus-ROI crop + warp + label bookkeeping
       + intensity augmentation on the GPU (train), deterministic crop (test loader)
    -> train.fit: 2 epochs, validation epoch after each (scripts/train_poseestimator.py:332-338 incl. the batch_idx-as-step quirk),
       CheckpointCallback = ModelCheckpoint(monitor="val_loss", filename="best", save_last=True) (:423-431)
    -> best.ckpt / last.ckpt in the plain save_model format, which the CPU oracle loads and evaluates to the same val_loss.
"""
import os
import shutil

import numpy as np
import pytest
import torch

from util import GOLDEN, train_script

pytestmark = pytest.mark.gpu


def _datadir(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, "aflw2kmini.npz"), d / "aflw2k.npz")  # the name the reference's constructor reads (aflw2k.h5)
    return str(d)


@pytest.mark.parametrize("graphed", [False, True])  # eager launches | every step a replay of one captured hipGraph (train.GraphedTrainStep)
def test_fit_from_shards_with_validation_and_best_checkpoint(tmp_path, monkeypatch, graphed):
    import trackertraincode.pipelines as P
    import trackertraincode.train as train
    from oracle import refmodel as R
    from trackertraincode.neuralnets.models import load_model
    from util import script_args

    S = train_script()
    datadir = _datadir(tmp_path)
    # the mini file has 16 frames: frames 0..7 validate, 8..15 train (the full set: 0..399 / 400..1999)
    monkeypatch.setitem(P._POSE_SHARDS, P.Id.AFLW2k3d, ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, 1000.0, (8, None)))
    monkeypatch.setattr(P, "_TEST_SHARD", ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, (0, 8)))
    ids, weights = S.parse_dataset_definition("aflw2k:500")
    assert ids == [P.Id.AFLW2k3d] and weights == {P.Id.AFLW2k3d: 500.0}
    with pytest.raises(ValueError):
        S.parse_dataset_definition("nosuchset")
    train_loader, test_loader, n = P.make_pose_estimation_loaders(129, 8, ids, dataset_weights=weights, device="cuda", seed=3, datadir=datadir, steps_per_epoch=3)
    assert n == 8 and len(train_loader) == 3 and len(test_loader) == 1
    first = next(iter(train_loader))
    assert isinstance(first, list) and first[0].meta.tag == P.Tag.POSE_WITH_LANDMARKS and first[0]["image"].shape == (8, 1, 129, 129)
    assert first[0]["image"].dtype == torch.float32 and -0.51 <= float(first[0]["image"].min()) and float(first[0]["image"].max()) <= 0.51
    assert set(("pose", "coord", "roi", "pt3d_68", "shapeparam", "coord_convention_id")) <= set(first[0].keys())
    vb = next(iter(test_loader))
    assert vb["image"].shape == (8, 1, 129, 129) and float(vb["coord"][:, 2].min()) > 0.2  # head sizes in crop units: faces fill the crop
    with pytest.raises(FileNotFoundError, match="h5_to_npz"):
        P.make_pose_estimation_loaders(129, 8, [P.Id.WFLW_LP], device="cuda", datadir=datadir)
    with pytest.raises(NotImplementedError):
        P.make_pose_estimation_loaders(129, 8, [P.Id.WIDER], device="cuda", datadir=datadir)

    args = script_args(dict(with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False), epochs=2)
    torch.manual_seed(0)
    net = S.create_net(args).cuda()
    g = torch.Generator().manual_seed(7)
    net.landmarks.deformablekeypoints.set_basis(torch.randn(68, 3, generator=g) * 0.5, torch.randn(50, 68, 3, generator=g) * 0.05)
    train_crit, test_crit = S.setup_losses(args, net)
    opt, sch = S.create_optimizer(net, args)
    out_dir = str(tmp_path / "ckpt")
    ck = train.CheckpointCallback(out_dir)
    seen = []

    class Spy:
        def on_validation_end(self, epoch, model, val_loss):
            seen.append((epoch, val_loss, model.training))

    train.fit(net, train_loader, train_crit, opt, sch, epochs=2, callbacks=[ck, Spy()], val_loader=test_loader, val_criterions=test_crit, graphed=graphed)
    assert [e for e, _, _ in seen] == [0, 1] and all(np.isfinite(v) and v > 0 for _, v, _ in seen) and all(t for _, _, t in seen)  # back in train mode
    assert ck.history == [v for _, v, _ in seen] and ck.best_value == min(ck.history) and ck.best_epoch == int(np.argmin(ck.history))
    assert os.path.exists(ck.best_model_path) and os.path.exists(ck.last_model_path)

    # the checkpoints are the plain format: load_model re-instantiates the network; the CPU oracle evaluates the loaded weights on the
    # validation crops to the same per-batch value validate() computed on the GPU
    last = load_model(ck.last_model_path)
    assert type(last).__name__ == "NetworkWithPointHead" and last.get_config() == net.get_config()
    for k, v in net.state_dict().items():
        assert torch.equal(last.state_dict()[k], v.cpu()), k
    val_gpu = train.validate(net, test_loader, test_crit)
    assert abs(val_gpu - ck.history[-1]) <= 1e-6 * abs(val_gpu)  # validation is deterministic and leaves the weights alone
    sd = {k: v.clone() for k, v in last.state_dict().items()}
    st = R.state_from_numpy({k: v.numpy() for k, v in sd.items()}, requires_grad=False)
    st["landmarks.deformablekeypoints.keypts"], st["landmarks.deformablekeypoints.keyeigvecs"] = sd["landmarks.deformablekeypoints.keypts"], sd["landmarks.deformablekeypoints.keyeigvecs"]
    gmm = R.ShapeGmm(os.path.join(GOLDEN, "shapeparams_gmm.npz"))
    ocrit, otest = R.setup_losses(with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False, epochs=2, gmm=gmm)
    total, count = 0.0, 0
    for bi, b in enumerate(test_loader):
        with torch.no_grad():
            o, _ = R.network_forward(st, b["image"].cpu(), None, net.get_config() | {"enable_point_head": True}, False)
        sub = {k: v.cpu() for k, v in b.items() if torch.is_tensor(v)}
        vals = [fn(o, sub) * (w(bi) if callable(w) else w) for _, fn, w in otest["POSE_WITH_LANDMARKS"]]
        total += float(torch.cat([v.reshape(-1) for v in vals]).sum()) * b.meta.batchsize
        count += b.meta.batchsize
    assert abs(total / count - val_gpu) <= 2e-3 * abs(val_gpu), (total / count, val_gpu)


def test_roi_override_extent_to_forehead(tmp_path, monkeypatch):
    """roi_override="extent_to_forehead" (reference pipelines.py:351-356): the crop is taken around the xy extent of the posed BFM head mesh.
    With a synthetic blob of the missing files' format (the arithmetic is pinned in tests/test_bfm.py); without any blob: FileNotFoundError."""
    import trackertraincode.pipelines as P
    from oracle.synth import write_synthetic_bfm_blob
    from trackertraincode.datatransformation.batch import head_extent_roi
    from trackertraincode.facemodel.bfm import BFMModel, ScaledBfmModule

    datadir = _datadir(tmp_path)
    monkeypatch.setitem(P._POSE_SHARDS, P.Id.AFLW2k3d, ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, 1000.0, (8, None)))
    monkeypatch.setattr(P, "_TEST_SHARD", ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, (0, 8)))
    have_blob = os.path.exists(os.path.join(os.path.dirname(P.__file__), "facemodel", "bfm_noneck_v3.pkl"))
    if not have_blob:
        with pytest.raises(FileNotFoundError, match="bfm_noneck_v3"):
            P.make_pose_estimation_loaders(129, 8, [P.Id.AFLW2k3d], device="cuda", datadir=datadir, roi_override="extent_to_forehead")
    blob = tmp_path / "bfm"
    blob.mkdir()
    write_synthetic_bfm_blob(str(blob))
    mesh = ScaledBfmModule(BFMModel(folder=str(blob)))
    _, test_o, _ = P.make_pose_estimation_loaders(129, 8, [P.Id.AFLW2k3d], device="cuda", datadir=datadir, enable_image_aug=False)
    _, test_h, _ = P.make_pose_estimation_loaders(129, 8, [P.Id.AFLW2k3d], device="cuda", datadir=datadir, enable_image_aug=False,
                                                  roi_override="extent_to_forehead", headmodel=mesh)
    fo, fh = test_o.datasets[0].fields, test_h.datasets[0].fields
    want = head_extent_roi(mesh.vertices.cuda(), fo["coord"], fo["pose"])
    assert torch.allclose(fh["roi"], want) and not torch.allclose(fh["roi"], fo["roi"])
    bo, bh = next(iter(test_o)), next(iter(test_h))
    # the synthetic mesh is a point cloud of about +-2 head sizes: its box is larger than the face box, so the same head fills less of the crop
    assert float((bh["coord"][:, 2] / bo["coord"][:, 2]).max()) < 1.0
