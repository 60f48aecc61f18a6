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
