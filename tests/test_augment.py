"""Affine-warp augmentation: oracle and host modules against the reference's goldens on
the CPU; the HIP kernels (through the C-ABI) against both on the GPU.  view_roi is INTEGER: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import augment as A
from util import GOLDEN

G = np.load(os.path.join(GOLDEN, "augment.npz"))
N = 129


def test_oracle_view_roi_known_answers_bit_exact():
    v = A.compute_view_roi(G["kat_bbox"], G["kat_f"], G["kat_t"])
    assert v.tolist() == G["kat_expected"].tolist()  # the 8 rows of test/test_affine_img_trafo.py:49-61
    vr = A.round_view_roi(A.compute_view_roi(G["roi"], G["scales"], G["translations"]))
    assert np.array_equal(vr, G["view_roi"])
    assert A.round_view_roi(np.array([0.5, 1.5, 2.5, -0.5], np.float32)).tolist() == [0, 2, 2, 0]  # half to even


def test_oracle_transforms_and_warp_match_reference():
    tr = A.crop_transform(G["view_roi"], G["angles"].astype(np.float64), N)
    np.testing.assert_allclose(tr, G["tr"], rtol=2e-5, atol=2e-4)
    nm = A.normalization(N)
    m = G["tr"]
    np.testing.assert_allclose(A.transform_coord(nm, A.transform_coord(m, G["coord"])), G["out_coord"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(A.transform_rot(nm, A.transform_rot(m, G["pose"])), G["out_pose"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(A.transform_roi(nm, A.transform_roi(m, G["roi"])), G["out_roi"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(A.transform_keypoints(nm, A.transform_keypoints(m, G["pt3d_68"])), G["out_pt3d_68"], rtol=1e-4, atol=1e-5)
    for b in (0, 3, 7):
        crop = A.warp_bilinear(G["image"][b, 0].astype(np.float64), G["tr"][b], N)
        np.testing.assert_allclose(crop, G["crop"][b, 0], atol=2e-2)  # grey levels 0..255
    f = G["flip_tr"]
    np.testing.assert_allclose(A.transform_keypoints(f, G["pt3d_68"][0]), G["flip_pt3d_68"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(A.transform_rot(f, G["pose"][0]), G["flip_pose"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(A.transform_coord(f, G["coord"][0]), G["flip_coord"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(A.transform_roi(f, G["roi"][0]), G["flip_roi"], rtol=1e-5, atol=1e-4)


def test_host_modules_match_reference():
    from trackertraincode.datatransformation.batch.geometric import GeneralFocusRoi, RoiFocusRandomizationParameters
    from trackertraincode.datatransformation.tensors.affinetrafo import FieldCategory, apply_affine2d, position_normalization
    from trackertraincode.neuralnets.affine2d import Affine2d

    t = torch.from_numpy
    v = GeneralFocusRoi._compute_view_roi(t(G["kat_bbox"]), t(G["kat_f"]), t(G["kat_t"]), 0.3)
    assert v.numpy().tolist() == G["kat_expected"].tolist()
    foc = GeneralFocusRoi(None, N)
    view, tr = foc.transform_for(t(G["roi"]), RoiFocusRandomizationParameters(t(G["scales"]), t(G["angles"]), t(G["translations"])))
    assert np.array_equal(view.numpy(), G["view_roi"])
    np.testing.assert_allclose(tr.tensor().numpy(), G["tr"], rtol=2e-5, atol=2e-4)
    nm, trg = position_normalization(N, N), Affine2d(t(G["tr"]))
    for key, cat in (("coord", FieldCategory.xys), ("pose", FieldCategory.quat), ("roi", FieldCategory.roi), ("pt3d_68", FieldCategory.points)):
        out = apply_affine2d(nm.expand(len(G[key])), key, apply_affine2d(trg, key, t(G[key]), cat), cat)
        np.testing.assert_allclose(out.numpy(), G["out_" + key], rtol=1e-4, atol=1e-5, err_msg=key)
    f = Affine2d(t(G["flip_tr"]))
    np.testing.assert_allclose(apply_affine2d(f, "pt3d_68", t(G["pt3d_68"][0]), FieldCategory.points).numpy(), G["flip_pt3d_68"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(apply_affine2d(f, "pose", t(G["pose"][0]), FieldCategory.quat).numpy(), G["flip_pose"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_augment_matches_reference_golden():
    from trackertraincode.datasets.batch import Batch, Metadata
    from trackertraincode.datatransformation import GpuFocusRoiAugment
    from trackertraincode.datatransformation.batch.geometric import RoiFocusRandomizationParameters

    dev = "cuda"
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    B = G["roi"].shape[0]
    batch = Batch(Metadata(96, B, tag="x"), image=t("image"), roi=t("roi"), coord=t("coord"), pose=t("pose"), pt3d_68=t("pt3d_68"))
    params = RoiFocusRandomizationParameters(t("scales"), t("angles"), t("translations"))
    out = GpuFocusRoiAugment(N, whiten=True)(batch, params=params)
    assert np.array_equal(out.view_roi.cpu().numpy(), G["view_roi"])  # INTEGER: bit-exact
    np.testing.assert_allclose(out.transform.cpu().numpy(), G["tr"], rtol=2e-5, atol=2e-4)
    crop = (out["image"].cpu().numpy() + 0.5) * 256.0
    assert out["image"].shape == (B, 1, N, N)
    np.testing.assert_allclose(crop, G["crop"], atol=5e-2)  # grey levels; fp32 gather arithmetic
    for k in ("coord", "pose", "roi", "pt3d_68"):
        np.testing.assert_allclose(out[k].cpu().numpy(), G["out_" + k], rtol=1e-4, atol=2e-5, err_msg=k)
    # known-answer rows through the kernel
    import trackertraincode._hip as H
    kat = torch.empty((8, 4), dtype=torch.int32, device=dev)
    kb, kf, kt = t("kat_bbox"), t("kat_f"), t("kat_t")  # keep the device tensors alive across the launch
    H.lib().call("ttk_view_roi", H.ptr(kb), H.ptr(kf), H.ptr(kt), 0.3, 8, H.ptr(kat))
    assert kat.cpu().numpy().tolist() == G["kat_expected"].astype(np.int32).tolist()
    # mirrored transform: flip map + sign handling
    ftr = t("flip_tr")[None].contiguous()
    pts_in, pts_out = t("pt3d_68")[:1].contiguous(), torch.empty((1, 68, 3), device=dev)
    pose, coord, roi = t("pose")[:1].clone(), t("coord")[:1].clone(), t("roi")[:1].clone()
    H.lib().call("ttk_affine_labels", H.ptr(ftr), 1, 0, H.ptr(coord), H.ptr(pose), H.ptr(roi), H.ptr(pts_in), H.ptr(pts_out))
    np.testing.assert_allclose(pts_out[0].cpu().numpy(), G["flip_pt3d_68"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(pose[0].cpu().numpy(), G["flip_pose"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(coord[0].cpu().numpy(), G["flip_coord"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(roi[0].cpu().numpy(), G["flip_roi"], rtol=1e-5, atol=1e-4)


GF = dict(np.load(os.path.join(GOLDEN, "augment_fliprot.npz"))) if os.path.exists(os.path.join(GOLDEN, "augment_fliprot.npz")) else None


def test_fliprot_table_matches_reference_composition():
    """The six (rot_dir, do_flip) point transforms of GpuFocusRoiAugment are exact quarter turns / mirrors of the 129-pixel crop (the
    reference composes them from range remaps and a rotation by +-pi/2, batch/geometric.py:241-251; cos(pi/2) leaves 4e-8)."""
    from trackertraincode.datatransformation import GpuFocusRoiAugment

    T = GpuFocusRoiAugment(N, flip_rot_p=0.01).fliprot_table().numpy()
    assert T.shape == (6, 3, 3)
    np.testing.assert_allclose(T[2], np.eye(3), atol=1e-6)                                   # rot 0, no flip
    np.testing.assert_allclose(T[3], [[-1, 0, N], [0, 1, 0], [0, 0, 1]], atol=1e-4)          # mirror x -> N - x
    for code in range(6):
        assert abs(abs(np.linalg.det(T[code][:2, :2])) - 1) < 1e-5
        assert (np.linalg.det(T[code][:2, :2]) < 0) == bool(code & 1)
    # the draw: mirrored half of the time, turned 1 % of the time
    aug = GpuFocusRoiAugment(N, flip_rot_p=0.01)
    codes = aug.draw_fliprot(200000, torch.Generator().manual_seed(3))
    assert abs(float((codes % 2 == 1).float().mean()) - 0.5) < 0.01
    assert abs(float((codes // 2 != 1).float().mean()) - 0.01) < 0.002 and abs(float((codes // 2 == 0).float().mean()) - 0.005) < 0.0015


@pytest.mark.gpu
@pytest.mark.parametrize("code", range(6))
def test_hip_flip_and_rot90_matches_reference_golden(code):
    """horizontal_flip_and_rot_90 behind the crop (reference batch/geometric.py:234-267, every one of its six draws forced in the golden):
    here the mirror / quarter turn is composed onto the crop's transform, so ONE warp produces the permuted crop and the labels see the
    composed transform."""
    from trackertraincode.datasets.batch import Batch, Metadata
    from trackertraincode.datatransformation import GpuFocusRoiAugment
    from trackertraincode.datatransformation.batch.geometric import RoiFocusRandomizationParameters

    dev = "cuda"
    b = int(GF["sample"])
    t = lambda k: torch.from_numpy(G[k][b:b + 1]).to(dev)
    batch = Batch(Metadata(96, 1, tag="x"), image=t("image"), roi=t("roi"), coord=t("coord"), pose=t("pose"), pt3d_68=t("pt3d_68"))
    params = RoiFocusRandomizationParameters(t("scales"), t("angles"), t("translations"))
    out = GpuFocusRoiAugment(N, whiten=True, flip_rot_p=0.01)(batch, params=params, fliprot_codes=torch.tensor([code]))
    crop = (out["image"].cpu().numpy()[0] + 0.5) * 256.0
    # the reference permutes the pixels of ITS crop; here the source is re-sampled at the permuted pixel centres: the same bilinear taps
    np.testing.assert_allclose(crop, GF[f"image_{code}"], atol=6e-2)
    if code == 2:
        np.testing.assert_allclose(crop, G["crop"][b], atol=5e-2)
    for k in ("coord", "pose", "roi", "pt3d_68"):
        np.testing.assert_allclose(out[k].cpu().numpy()[0], GF[f"{k}_{code}"], rtol=1e-4, atol=3e-5, err_msg=f"{k} code {code}")


@pytest.mark.gpu
def test_hip_roi_from_landmarks_matches_reference_golden():
    """roi_override="landmarks" (pipelines.py:343-350): PutRoiFromLandmarks in front of the deterministic 1.2x crop and behind it."""
    from trackertraincode.datasets.batch import Batch, Metadata
    from trackertraincode.datatransformation import GpuFocusRoiAugment
    from trackertraincode.datatransformation.batch.geometric import NoRoiRandomization

    dev = "cuda"
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    B = G["roi"].shape[0]
    batch = Batch(Metadata(96, B, tag="x"), image=t("image"), roi=t("roi"), coord=t("coord"), pose=t("pose"), pt3d_68=t("pt3d_68"))
    out = GpuFocusRoiAugment(N, whiten=True, make_params=NoRoiRandomization(1.2), roi_from_landmarks=True)(batch)
    assert np.array_equal(out.view_roi.cpu().numpy(), GF["lm_view_roi"])  # INTEGER: bit-exact
    np.testing.assert_allclose(out.transform.cpu().numpy(), GF["lm_tr"], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(out["roi"].cpu().numpy(), GF["lm_roi_after"], rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(out["pt3d_68"].cpu().numpy(), GF["lm_pt3d_68_after"], rtol=1e-4, atol=3e-5)
    # the box in front of the crop: xy extent of the landmarks
    xy = G["pt3d_68"][..., :2]
    np.testing.assert_allclose(np.concatenate([xy.min(1), xy.max(1)], -1), GF["lm_roi_before"], rtol=0, atol=0)
