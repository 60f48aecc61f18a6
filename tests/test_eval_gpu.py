"""Evaluation path (SURVEY.md §8 f1 + the accuracy harness of §8d): AFLW2000-3D style pose MAE on the bundled
aflw2kmini crops, HIP `Predictor` (GPU crop + eval-mode network + back-transformation) against the CPU oracle
running the same pipeline in numpy/torch-CPU on identical weights and pixels.  north_star: rotation MAE within
0.05 degrees of the reference path."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import augment as A
from oracle import refmodel as R
from oracle.synth import make_state
from util import GOLDEN, build_net, load_golden

pytestmark = pytest.mark.gpu


def _load_mini():
    from PIL import Image

    d = np.load(os.path.join(GOLDEN, "aflw2kmini.npz"))
    off, images = 0, []
    for n in d["image_lengths"]:
        img = Image.open(io.BytesIO(d["image_bytes"][off:off + int(n)].tobytes()))
        images.append(np.array(img))
        off += int(n)
    return images, d


def _grey(img):
    if img.ndim == 3 and img.shape[-1] == 3:
        return np.clip(np.rint((img.astype(np.float32) * np.array([0.299, 0.587, 0.114], np.float32)).sum(-1)), 0, 255).astype(np.uint8)
    return img if img.ndim == 2 else img[..., 0]


def _euler_deg(q):
    from trackertraincode import utils
    return np.array([utils.inv_aflw_rotation_conversion(r) for r in utils.convert_to_rot(q)]) * utils.rad2deg


@pytest.mark.parametrize("cfg", ["default", "full"])
def test_aflw2kmini_pose_mae_parity(cfg):
    from trackertraincode import eval as E

    images, d = _load_mini()
    rois = d["rois"].astype(np.float32)
    g, meta = load_golden(f"model_{cfg}.npz")
    cal = {k[len("calib/"):]: g[k] for k in g.files if k.startswith("calib/")}
    net = build_net(meta, "cuda", cal).eval()
    N = net.input_resolution

    # ---- HIP path
    pred = E.Predictor(net)
    t_images = [torch.from_numpy(im) for im in images]
    out = pred.predict_batch(t_images, torch.from_numpy(rois))
    targets = {"pose": torch.from_numpy(d["quats"].astype(np.float32)).cuda(), "roi": torch.from_numpy(rois).cuda(),
               "coord": torch.from_numpy(d["coords"].astype(np.float32)).cuda()}
    m_euler, m_geo = E.EulerAngleErrors(), E.GeodesicError()
    m_euler.update(out, targets)
    m_geo.update(out, targets)
    table = E.pose_error_table(m_euler.compute(), m_geo.compute())

    # ---- oracle path: same crop arithmetic in numpy, oracle network, same back-transformation
    view = A.round_view_roi(A.compute_view_roi(rois, np.full(len(rois), 1.1, np.float32), np.zeros((len(rois), 2), np.float32)))
    crops, mats = [], []
    for im, v in zip(images, view):
        m = A.crop_transform(v, 0.0, N)
        crops.append(A.warp_bilinear(_grey(im).astype(np.float32), m, N) / 256.0 - 0.5)
        mats.append(m)
    x = torch.from_numpy(np.stack(crops)[:, None].astype(np.float32))
    sd = make_state({k: tuple(v) for k, v in meta["shapes"].items()}, meta["state_seed"])
    sd.update(cal)
    st = R.state_from_numpy(sd, requires_grad=False)
    with torch.no_grad():
        ref, _ = R.network_forward(st, x, None, meta["config"], False)
    back = []
    for m in mats:
        full = np.vstack([A.normalization(N).astype(np.float64) @ np.vstack([m.astype(np.float64), [0, 0, 1]]), [0, 0, 1]])
        back.append(np.linalg.inv(full)[:2].astype(np.float32))
    back = np.stack(back)
    ref_pose = A.transform_rot(back, ref["pose"].numpy())
    ref_coord = A.transform_coord(back, ref["coord"].numpy())

    # crops agree (bilinear resampling of identical pixels)
    crop_hip = pred.crop_batch(t_images, torch.from_numpy(rois))["image"].cpu().numpy()[:, 0]
    np.testing.assert_allclose(crop_hip, np.stack(crops), atol=2e-3)
    # per-sample predictions agree
    q_hip = out["pose"].cpu().numpy()
    sign = np.sign((q_hip * ref_pose).sum(-1, keepdims=True))
    np.testing.assert_allclose(q_hip * sign, ref_pose, atol=2e-4)
    np.testing.assert_allclose(out["coord"].cpu().numpy(), ref_coord, rtol=1e-3, atol=5e-2)  # pixels
    # the headline number: Euler-angle MAE of both paths against the labels
    e_hip = np.abs(m_euler.compute().cpu().numpy()) * 180.0 / np.pi
    e_ref = E._angle_errors(_euler_deg(ref_pose) * np.pi / 180.0, _euler_deg(d["quats"]) * np.pi / 180.0) * 180.0 / np.pi
    assert np.abs(e_hip - e_ref).max() < 0.05
    assert abs(e_hip.mean() - e_ref.mean()) < 0.05 and abs(table["mae"] - e_ref.mean()) < 0.05
    if "pt3d_68" in out:
        ref_pts = A.transform_points(back, ref["pt3d_68"].numpy())
        np.testing.assert_allclose(out["pt3d_68"].cpu().numpy(), ref_pts, rtol=1e-3, atol=5e-2)


def test_aflw2kmini_training_step_matches_oracle():
    """BASELINE config 1: one fwd+bwd step of the default pose estimator on the bundled aflw2kmini samples - here on the
    MI355X path (GPU crop + label bookkeeping, HIP network, HIP losses) against the CPU oracle on the same pixels."""
    import trackertraincode.train as train
    from trackertraincode.datasets.batch import Batch, Metadata
    from trackertraincode.datatransformation.batch.geometric import NoRoiRandomization
    from trackertraincode.datatransformation.gpu import GpuFocusRoiAugment
    from trackertraincode.pipelines import Tag
    from util import script_args, train_script

    images, d = _load_mini()
    sizes = {im.shape[:2] for im in images}
    assert len(sizes) == 1, "aflw2kmini images share one size"
    B = len(images)
    g, meta = load_golden("model_default.npz")
    net = build_net(meta, "cuda").train()
    S = train_script()
    crit, _ = S.setup_losses(script_args(meta["flags"]), net)
    N = net.input_resolution
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    grey = np.stack([_grey(im) for im in images])
    lab = {"coord": d["coords"].astype(np.float32), "pose": d["quats"].astype(np.float32), "roi": d["rois"].astype(np.float32),
           "pt3d_68": d["pt3d_68"].astype(np.float32), "shapeparam": d["shapeparams"].astype(np.float32)}
    raw = Batch(Metadata(grey.shape[1:][::-1], B, tag=Tag.POSE_WITH_LANDMARKS), image=t(grey[:, None]), **{k: t(v) for k, v in lab.items()},
                coord_convention_id=torch.zeros(B, dtype=torch.int32, device="cuda"))
    batch = GpuFocusRoiAugment(new_size=N, make_params=NoRoiRandomization(1.1))(raw)
    out = train.training_step(net, [batch], 0, crit)
    out["loss"].backward()

    # oracle: numpy crop + label transforms, oracle network + losses
    view = A.round_view_roi(A.compute_view_roi(lab["roi"], np.full(B, 1.1, np.float32), np.zeros((B, 2), np.float32)))
    mats = np.stack([A.crop_transform(v, 0.0, N) for v in view])
    crops = np.stack([A.warp_bilinear(gi.astype(np.float32), m, N) / 256.0 - 0.5 for gi, m in zip(grey, mats)])
    full = np.stack([(A.normalization(N).astype(np.float64) @ np.vstack([m.astype(np.float64), [0, 0, 1]])).astype(np.float32) for m in mats])
    olab = {"coord": A.transform_coord(full, lab["coord"]), "pose": A.transform_rot(full, lab["pose"]), "roi": A.transform_roi(full, lab["roi"]),
            "pt3d_68": A.transform_keypoints(full, lab["pt3d_68"]), "shapeparam": lab["shapeparam"]}
    for k in ("coord", "pose", "roi", "pt3d_68"):
        np.testing.assert_allclose(batch[k].cpu().numpy(), olab[k], rtol=1e-4, atol=1e-4, err_msg=k)
    sd = make_state({k: tuple(v) for k, v in meta["shapes"].items()}, meta["state_seed"])
    st = R.state_from_numpy(sd)
    gmm = R.ShapeGmm(os.path.join(GOLDEN, "shapeparams_gmm.npz"))
    fl = meta["flags"]
    ocrit, _ = R.setup_losses(with_pointhead=fl["with_pointhead"], with_nll_loss=fl["with_nll_loss"],
                              rampup_nll_losses=fl["rampup_nll_losses"], epochs=200, gmm=gmm)
    pred, _ = R.network_forward(st, torch.from_numpy(crops[:, None].astype(np.float32)), torch.zeros(B, dtype=torch.int64), meta["config"], True)
    ob = dict(tag="POSE_WITH_LANDMARKS", n=B, **{k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in olab.items()})
    loss_ref, by_name = R.compute_loss(pred, [ob], 0, ocrit)
    assert abs(out["loss"].item() - loss_ref.item()) < 1e-3  # north_star: per-step losses within 1e-3
    for k, (v, _) in by_name.items():
        np.testing.assert_allclose(out["mt_losses"][k].detach().cpu().numpy(), v.detach().numpy(), rtol=1e-3, atol=1e-3, err_msg=k)
