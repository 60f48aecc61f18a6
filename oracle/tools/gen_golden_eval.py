#!/usr/bin/env python3
"""Golden vectors of the EVALUATION path and of the two data-side transforms that round 4 builds (row f2), produced by
the REFERENCE's own functions imported through ref_shims (build container only; /root/reference does not travel):

  eval.py:  _angle_errors :340-344, _quat_to_aflw3d_rotations / _aflw3d_euler_errors :347-358 (utils.convert_to_rot,
            utils.inv_aflw_rotation_conversion), GeodesicError :335-337 (torchquaternion.geodesicdistance), NormalizedXYSError :366-371,
            _eval_keypoints :375-391 (dims 2 and 3), KptNME._compute_bin_masks :423-437 and its bin means :417-421,
            Predictor's way back to image coordinates :199-206 (batch/normalization.py unnormalize_batch, eval._apply_backtrafo with the
            image_backtransform that FocusRoi(insert_backtransform=True) stores: tensors/affinetrafo.py:19-34)
  datatransformation/batch/geometric.py:234-267  horizontal_flip_and_rot_90 for all six (rot_dir, do_flip) draws (np.random patched to
            force each) on a 129 x 129 crop with every label category
  datatransformation/batch/misc.py:9-31          PutRoiFromLandmarks(extend_to_forehead=False) in front of and behind the deterministic crop
            (pipelines.py:343-350, roi_override="landmarks", extension factor 1.2)

-> tests/golden/eval.npz, tests/golden/augment_fliprot.npz  (inputs are stored too)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shims  # noqa: E402

torch = ref_shims.install(synthetic_bfm=True)
import trackertraincode.eval as RE  # noqa: E402
from trackertraincode import utils as RU  # noqa: E402
from trackertraincode.datasets.batch import Batch, Metadata  # noqa: E402
from trackertraincode.datasets.dshdf5pose import FieldCategory  # noqa: E402
import trackertraincode.datatransformation as dtr  # noqa: E402
from trackertraincode.datatransformation.batch import geometric as G  # noqa: E402
from trackertraincode.datatransformation.batch.geometric import GeneralFocusRoi  # noqa: E402
from trackertraincode.datatransformation.tensors.affinetrafo import apply_affine2d, position_normalization  # noqa: E402
from trackertraincode.neuralnets import torchquaternion  # noqa: E402
from trackertraincode.neuralnets.affine2d import Affine2d  # noqa: E402

rng = np.random.default_rng(404)
out = {}
# ---------------------------------------------------------------- metrics
B = 48
q1 = rng.standard_normal((B, 4)).astype(np.float32)
q1 /= np.linalg.norm(q1, axis=-1, keepdims=True)
dq = np.concatenate([rng.standard_normal((B, 3)) * 0.08, np.ones((B, 1))], -1).astype(np.float32)
dq /= np.linalg.norm(dq, axis=-1, keepdims=True)
q2 = torchquaternion.mult(torch.from_numpy(q1), torch.from_numpy(dq)).numpy().astype(np.float32)
q1[:6, 3] = np.abs(q1[:6, 3]) + 2.0  # a few near-frontal poses (all yaw bins get members below)
q1 /= np.linalg.norm(q1, axis=-1, keepdims=True)
e1, e2 = rng.uniform(-np.pi, np.pi, (B, 3)), rng.uniform(-np.pi, np.pi, (B, 3))
out.update(q1=q1, q2=q2, euler1=e1, euler2=e2)
out["angle_errors"] = RE._angle_errors(e1, e2)
out["aflw3d_rotations_q1"] = RE._quat_to_aflw3d_rotations(torch.from_numpy(q1))
out["aflw3d_euler_errors"] = RE._aflw3d_euler_errors(torch.from_numpy(q1), torch.from_numpy(q2)).numpy()
out["geodesic"] = torchquaternion.geodesicdistance(torch.from_numpy(q2), torch.from_numpy(q1)).numpy()  # GeodesicError: (targets, preds)
coord_p = np.stack([rng.uniform(100, 300, B), rng.uniform(100, 300, B), rng.uniform(40, 90, B)], -1).astype(np.float32)
coord_t = (coord_p + rng.standard_normal((B, 3)) * 3).astype(np.float32)
roi_t = np.stack([coord_t[:, 0] - 60, coord_t[:, 1] - 70, coord_t[:, 0] + 65, coord_t[:, 1] + 75], -1).astype(np.float32)
out.update(coord_pred=coord_p, coord_target=coord_t, roi_target=roi_t)
out["normalized_xys"] = RE.NormalizedXYSError.compute_on_batch(None, {"coord": torch.from_numpy(coord_p)},
                                                              {"coord": torch.from_numpy(coord_t), "roi": torch.from_numpy(roi_t)}).numpy()
pts_t = np.concatenate([rng.uniform(120, 280, (B, 68, 2)), rng.uniform(-40, 40, (B, 68, 1))], -1).astype(np.float32)
pts_p = (pts_t + rng.standard_normal((B, 68, 3)) * 2.5).astype(np.float32)
out.update(pts_pred=pts_p, pts_target=pts_t)
for dims in (2, 3):
    out[f"kpt_nme_{dims}d"] = RE._eval_keypoints(torch.from_numpy(pts_p), torch.from_numpy(pts_t), dims).numpy()
masks = RE.KptNME._compute_bin_masks(None, torch.from_numpy(q1))
out["yaw_bin_masks"] = masks.numpy()
errs = torch.from_numpy(out["kpt_nme_3d"])
out["kpt_nme_bins"] = np.array([torch.mean(errs[m]).item() for m in masks.unbind(-1)], np.float64)  # KptNME.compute :417-421

# ---------------------------------------------------------------- Predictor: crop coordinates -> image coordinates
N, ext = 129, 1.1
Bp = 10
rois = np.stack([rng.uniform(40, 90, Bp), rng.uniform(30, 80, Bp), rng.uniform(200, 330, Bp), rng.uniform(210, 350, Bp)], -1).astype(np.float32)
foc = GeneralFocusRoi(None, N, "roi", True)
pred = {"coord": np.stack([rng.uniform(-0.3, 0.3, Bp), rng.uniform(-0.3, 0.3, Bp), rng.uniform(0.3, 0.7, Bp)], -1).astype(np.float32),
        "pose": (lambda q: (q / np.linalg.norm(q, axis=-1, keepdims=True)).astype(np.float32))(rng.standard_normal((Bp, 4))),
        "pt3d_68": rng.uniform(-0.8, 0.8, (Bp, 68, 3)).astype(np.float32),
        "roi": np.stack([rng.uniform(-0.9, -0.4, Bp), rng.uniform(-0.9, -0.4, Bp), rng.uniform(0.4, 0.9, Bp), rng.uniform(0.4, 0.9, Bp)], -1).astype(np.float32)}
cats = {"coord": FieldCategory.xys, "pose": FieldCategory.quat, "pt3d_68": FieldCategory.points, "roi": FieldCategory.roi}
back, crop_tr = {k: [] for k in pred}, []
for b in range(Bp):
    # FocusRoi(N, 1.1, insert_backtransform=True): NoRoiRandomization -> scale 1.1, no shift, no rotation (geometric.py:52-56,87-97,193-224)
    vr = GeneralFocusRoi._compute_view_roi(torch.from_numpy(rois[b]), torch.tensor(ext), torch.zeros(2), 0.3)
    vr = torch.round(vr).to(torch.int32)
    tr = foc._compute_point_transform_from_roi((), vr, N)
    crop_tr.append(tr.tensor().numpy())
    sample = Batch(Metadata(N, 0, categories=dict(cats)), {k: torch.from_numpy(v[b].copy()) for k, v in pred.items()})
    # predict_batch :186-198: FocusRoi stores tr.inv() (affinetrafo.py:26), normalize_batch carries it along (apply_affine2d's
    # "image_backtransform" rule, :140-147), the predictions inherit it, unnormalize_batch undoes the normalisation on both
    crop_side = Batch(Metadata(N, 0), {"image_backtransform": tr.inv().tensor()})
    sample["image_backtransform"] = dtr.batch.normalize_batch(crop_side)["image_backtransform"]
    sample = dtr.batch.unnormalize_batch(sample)
    sample = RE._apply_backtrafo(Affine2d(sample.pop("image_backtransform")), sample)
    for k in pred:
        back[k].append(sample[k].numpy())
out.update(pred_rois=rois, crop_transform=np.stack(crop_tr).astype(np.float32))
for k, v in pred.items():
    out["pred_" + k] = v
    out["back_" + k] = np.stack(back[k]).astype(np.float32)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "eval.npz"), **out)
print("eval.npz", {k: v.shape for k, v in out.items()})

# ---------------------------------------------------------------- flip / rot90 behind the crop, all six draws
out = {}
aug = np.load(os.path.join(REPO, "tests", "golden", "augment.npz"))  # its sample 3: source image, crop transform, crop, labels
b = 3
tr = Affine2d(torch.from_numpy(aug["tr"][b]))
crop = torch.from_numpy(aug["crop"][b])  # [1, N, N] grey levels
lab_in = {"coord": (aug["coord"][b], FieldCategory.xys), "pose": (aug["pose"][b], FieldCategory.quat), "roi": (aug["roi"][b], FieldCategory.roi),
          "pt3d_68": (aug["pt3d_68"][b], FieldCategory.points)}
out["sample"] = np.int32(b)
norm = position_normalization(N, N)
for rot_dir in (-1, 0, 1):
    for do_flip in (0, 1):
        sample = Batch(Metadata(N, 0, categories={"image": FieldCategory.image, **{k: c for k, (_, c) in lab_in.items()}}),
                       {"image": crop.clone(), **{k: apply_affine2d(tr, k, torch.from_numpy(v.copy()), c) for k, (v, c) in lab_in.items()}})
        # force the two draws of :236-237
        G.np.random.randint = lambda lo, hi, _f=do_flip: 0 if _f else 1
        G.np.random.choice = lambda a, p=None, _r=rot_dir: _r
        res = G.horizontal_flip_and_rot_90(0.01, sample)
        code = (rot_dir + 1) * 2 + do_flip
        out[f"image_{code}"] = res["image"].numpy().astype(np.float32)
        for k, (_, c) in lab_in.items():
            out[f"{k}_{code}"] = apply_affine2d(norm, k, res[k], c).numpy().astype(np.float32)  # normalize_batch follows (pipelines.py:377)
np.random.seed(0)

# ---------------------------------------------------------------- roi_override="landmarks": PutRoiFromLandmarks . FocusRoi(N, 1.2) . PutRoiFromLandmarks
# (the constructor builds the BFM head model, whose blob the reference's repository lacks; extend_to_forehead=False never touches it, so
# the object is made without running __init__ and its own __call__ / _create_roi do the work)
put = object.__new__(dtr.batch.PutRoiFromLandmarks)
put.extend_to_forehead = False
Bl = aug["pt3d_68"].shape[0]
roi0, view, trs, roi1, pts1 = [], [], [], [], []
for i in range(Bl):
    s = Batch(Metadata(96, 0, categories={"pt3d_68": FieldCategory.points, "roi": FieldCategory.roi}),
              {"pt3d_68": torch.from_numpy(aug["pt3d_68"][i].copy()), "roi": torch.from_numpy(aug["roi"][i].copy())})
    s = put(s)
    roi0.append(s["roi"].numpy())
    vr = torch.round(GeneralFocusRoi._compute_view_roi(s["roi"], torch.tensor(1.2), torch.zeros(2), 0.3)).to(torch.int32)
    t = foc._compute_point_transform_from_roi((), vr, N)
    view.append(vr.numpy())
    trs.append(t.tensor().numpy())
    s2 = Batch(Metadata(N, 0, categories=dict(s.meta.categories)), {k: apply_affine2d(t, k, v, s.get_category(k)) for k, v in s.items()})
    s2 = put(s2)
    roi1.append(apply_affine2d(norm, "roi", s2["roi"], FieldCategory.roi).numpy())
    pts1.append(apply_affine2d(norm, "pt3d_68", s2["pt3d_68"], FieldCategory.points).numpy())
out.update(lm_roi_before=np.stack(roi0).astype(np.float32), lm_view_roi=np.stack(view).astype(np.int32), lm_tr=np.stack(trs).astype(np.float32),
           lm_roi_after=np.stack(roi1).astype(np.float32), lm_pt3d_68_after=np.stack(pts1).astype(np.float32))
np.savez_compressed(os.path.join(REPO, "tests", "golden", "augment_fliprot.npz"), **out)
print("augment_fliprot.npz", {k: v.shape for k, v in out.items()})
