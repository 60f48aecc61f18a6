#!/usr/bin/env python3
"""Golden vectors for the affine-warp augmentation, produced by the REFERENCE's own
functions imported through ref_shims (build container only):

  GeneralFocusRoi._compute_view_roi + round-to-int32      datatransformation/batch/geometric.py:108-157,205
  _compute_point_transform_from_roi / _center_rotation_tr  :159-177
  apply_affine2d for xys / quat / roi / points             datatransformation/tensors/affinetrafo.py:37-148
  position_normalization (normalize_batch)                 datatransformation/batch/normalization.py:20-56
  affine_transform_image_torch (bilinear, zeros, align_corners=False)   tensors/image_geometric_torch.py:60-98

-> tests/golden/augment.npz  (inputs are stored too: they are small)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shims  # noqa: E402

torch = ref_shims.install()
from trackertraincode.datasets.dshdf5pose import FieldCategory  # noqa: E402
from trackertraincode.datatransformation.batch.geometric import GeneralFocusRoi  # noqa: E402
from trackertraincode.datatransformation.tensors.affinetrafo import apply_affine2d, position_normalization  # noqa: E402
from trackertraincode.datatransformation.tensors.image_geometric_torch import affine_transform_image_torch  # noqa: E402
from trackertraincode.neuralnets.affine2d import Affine2d  # noqa: E402

rng = np.random.default_rng(2024)
B, S, N = 12, 96, 129  # source images S x S, crops N x N
out = {}
# ---- inputs
roi = np.stack([rng.uniform(10, 30, B), rng.uniform(10, 30, B), rng.uniform(60, 90, B), rng.uniform(55, 90, B)], -1).astype(np.float32)
scales = np.clip(rng.standard_normal(B) * 0.1, -0.5, 0.5).astype(np.float32) + np.float32(1.1)
transl = np.clip(rng.standard_normal((B, 2)) * 0.5, -1, 1).astype(np.float32)
angles = (np.pi * 30 / 180 * rng.choice([-1.0, 0.0, 1.0], B)).astype(np.float32)
angles[0] = 0.0
image = rng.integers(0, 256, (B, 1, S, S)).astype(np.uint8)
yy, xx = np.mgrid[0:S, 0:S]
image = (0.5 * image + 0.5 * (127 + 120 * np.sin(xx / 7.0 + np.arange(B)[:, None, None, None]) * np.cos(yy / 5.0))).clip(0, 255).astype(np.uint8)
coord = np.stack([rng.uniform(30, 60, B), rng.uniform(30, 60, B), rng.uniform(15, 30, B)], -1).astype(np.float32)
pose = rng.standard_normal((B, 4)).astype(np.float32)
pose /= np.linalg.norm(pose, axis=-1, keepdims=True)
pts = np.concatenate([rng.uniform(10, 90, (B, 68, 2)), rng.uniform(-20, 20, (B, 68, 1))], -1).astype(np.float32)
out.update(roi=roi, scales=scales, translations=transl, angles=angles, image=image, coord=coord, pose=pose, pt3d_68=pts)

# ---- the 8 integer known-answer rows of test/test_affine_img_trafo.py:49-61 (data) re-evaluated here
kat_in = [([-10, -10, 10, 10], 1.0, [-1.0, 0.0]), ([-10, -10, 10, 10], 1.0, [1.0, 0.0]), ([-10, -10, 10, 10], 1.0, [0.0, -1.0]),
          ([-10, -10, 10, 10], 1.0, [0.0, 1.0]), ([-10, -10, 10, 10], 2.0, [0.0, 0.0]), ([-10, -10, 10, 10], 2.0, [-1.0, 0.0]),
          ([-10, -10, 10, 10], 0.5, [0.0, 0.0]), ([-10, -10, 10, 10], 0.5, [-1.0, 0.0])]
kat = [GeneralFocusRoi._compute_view_roi(torch.tensor(b, dtype=torch.float32), torch.tensor(f), torch.tensor(t), 0.3).numpy() for b, f, t in kat_in]
out["kat_bbox"] = np.array([b for b, _, _ in kat_in], np.float32)
out["kat_f"] = np.array([f for _, f, _ in kat_in], np.float32)
out["kat_t"] = np.array([t for _, _, t in kat_in], np.float32)
out["kat_expected"] = np.array(kat, np.float32)
assert out["kat_expected"].tolist() == [[-16, -10, 4, 10], [-4, -10, 16, 10], [-10, -16, 10, 4], [-10, -4, 10, 16], [-20, -20, 20, 20],
                                        [-36, -20, 4, 20], [-5, -5, 5, 5], [-13, -5, -3, 5]]

# ---- per-sample pipeline exactly as GeneralFocusRoi.__call__ (:193-224) + normalize_batch, with the torch image warp
foc = GeneralFocusRoi(None, N, "roi", False)
view_rois, trs, crops, lab = [], [], [], {k: [] for k in ("coord", "pose", "roi", "pt3d_68")}
for b in range(B):
    vr = GeneralFocusRoi._compute_view_roi(torch.from_numpy(roi[b]), torch.tensor(scales[b]), torch.from_numpy(transl[b]), 0.3)
    vr = torch.round(vr).to(torch.int32)
    tr = foc._compute_point_transform_from_roi((), vr, N)
    tr = foc._center_rotation_tr(torch.tensor(angles[b])) @ tr
    view_rois.append(vr.numpy())
    trs.append(tr.tensor().numpy())
    crops.append(affine_transform_image_torch(torch.from_numpy(image[b]).float(), tr, N).numpy())
    norm = position_normalization(N, N)
    for key, val, cat in (("coord", coord[b], FieldCategory.xys), ("pose", pose[b], FieldCategory.quat),
                          ("roi", roi[b], FieldCategory.roi), ("pt3d_68", pts[b], FieldCategory.points)):
        v = apply_affine2d(tr, key, torch.from_numpy(val.copy()), cat)
        v = apply_affine2d(norm, key, v, cat)
        lab[key].append(v.numpy())
out["view_roi"] = np.stack(view_rois).astype(np.int32)
out["tr"] = np.stack(trs).astype(np.float32)
out["crop"] = np.stack(crops).astype(np.float32)  # float grey levels 0..255 before /256 and whitening
for k, v in lab.items():
    out["out_" + k] = np.stack(v).astype(np.float32)

# ---- a reflecting transform exercises the flip-map / det<0 branches of transform_keypoints / transform_rot
trf = Affine2d.horizontal_flip(torch.tensor(48.0)) @ Affine2d.trs(translations=torch.tensor([3.0, -2.0]), angles=torch.tensor(0.3), scales=torch.tensor(1.2))
out["flip_tr"] = trf.tensor().numpy()
out["flip_pt3d_68"] = apply_affine2d(trf, "pt3d_68", torch.from_numpy(pts[0].copy()), FieldCategory.points).numpy()
out["flip_pose"] = apply_affine2d(trf, "pose", torch.from_numpy(pose[0].copy()), FieldCategory.quat).numpy()
out["flip_coord"] = apply_affine2d(trf, "coord", torch.from_numpy(coord[0].copy()), FieldCategory.xys).numpy()
out["flip_roi"] = apply_affine2d(trf, "roi", torch.from_numpy(roi[0].copy()), FieldCategory.roi).numpy()
np.savez_compressed(os.path.join(REPO, "tests", "golden", "augment.npz"), **out)
print({k: v.shape for k, v in out.items()})
