"""Evaluation path: `Predictor` (face box -> 129x129 crop -> network -> predictions in image coordinates) and the pose /
landmark error metrics (reference: trackertraincode/eval.py:158-252, 295-440; scripts/evaluate_pose_network.py).

The crop is produced on the MI355X by the affine-warp kernels of csrc/warp.hip (the same kernels as the training
augmentation, with the deterministic parameters of the reference's `FocusRoi`: enlargement 1.1, no shift, no rotation),
the network runs its eval-mode HIP path, and the back-transformation to image coordinates is the inverse of the crop
transform applied with `apply_affine2d`.  The reference crops with OpenCV (`croprescale_image_cv2`, area / linear
filters); this build resamples bilinearly like its `affine_transform_image_torch` - predictions on real images differ
from an OpenCV crop at the level of the resampling filter, which is why MAE parity is asserted against the CPU
oracle running the SAME resampling (tests/test_eval_gpu.py), not against stored reference numbers.
torchmetrics is not a dependency here: the metrics are small accumulators with the reference's names and formulas."""
from __future__ import annotations

from typing import Dict, List, NamedTuple

import numpy as np
import torch
from torch import Tensor

from . import _hip, utils
from .datasets.batch import Batch, Metadata
from .datatransformation.batch.geometric import NoRoiRandomization
from .datatransformation.gpu import GpuFocusRoiAugment
from .datatransformation.tensors.affinetrafo import FieldCategory, apply_affine2d, position_normalization
from .neuralnets import torchquaternion
from .neuralnets.affine2d import Affine2d


class Predictor:
    """`predict_batch(images, rois)`: images = list of uint8 tensors [H,W] / [H,W,1] / [H,W,3] (grey levels or RGB,
    any sizes) or one [B,1,H,W] tensor; rois [B,4] = (x0, y0, x1, y1) face boxes in pixels.  Returns a Batch with the
    network outputs mapped back to image coordinates: coord [B,3] (x, y, head size in pixels), pose [B,4], roi [B,4],
    pt3d_68 [B,68,3] (when the network has the landmark head) - reference :185-208."""

    def __init__(self, net: torch.nn.Module, focus_roi_expansion_factor: float = 1.1, device: str | torch.device = "cuda"):
        self._net = net.to(device).eval()
        self._device = torch.device(device)
        self._crop = GpuFocusRoiAugment(new_size=net.input_resolution, make_params=NoRoiRandomization(focus_roi_expansion_factor))

    @property
    def input_resolution(self) -> int:
        return self._net.input_resolution

    @staticmethod
    def _grey(img: Tensor) -> Tensor:
        if img.dim() == 3 and img.shape[-1] == 3:  # ITU-R 601 luma, as the training data is stored (grey 8 bit)
            w = torch.tensor([0.299, 0.587, 0.114], dtype=torch.float32, device=img.device)
            img = (img.to(torch.float32) * w).sum(-1).round().clamp(0, 255).to(torch.uint8)
        elif img.dim() == 3:
            img = img[..., 0]
        return img

    def crop_batch(self, images, rois: Tensor) -> Batch:
        """Normalised crops [B,1,N,N] (whitened) + the pixel transforms image -> crop used for the back-transformation."""
        rois = rois.to(self._device, torch.float32)
        if torch.is_tensor(images):
            groups = {tuple(images.shape[-2:]): list(range(images.shape[0]))}
            get = lambda i: images[i, 0]
        else:
            groups: Dict[tuple, List[int]] = {}
            for i, im in enumerate(images):
                groups.setdefault(tuple(self._grey(im).shape), []).append(i)
            get = lambda i: self._grey(images[i])
        B = rois.shape[0]
        N = self.input_resolution
        crops = torch.empty((B, 1, N, N), dtype=torch.float32, device=self._device)
        trs = torch.empty((B, 2, 3), dtype=torch.float32, device=self._device)
        for _, idx in groups.items():  # images of one size are warped together
            src = torch.stack([get(i).to(self._device) for i in idx])[:, None].contiguous()
            sub = Batch(Metadata(tuple(src.shape[-2:][::-1]), len(idx)), image=src, roi=rois[idx])
            out = self._crop(sub)
            crops[idx], trs[idx] = out["image"], out.transform
        return Batch(Metadata(N, B), image=crops, image_transform=trs)

    @torch.no_grad()
    def predict_batch(self, images, rois: Tensor) -> Batch:
        if self._device.type == "cuda":
            _hip.lib().clear_stale_error("the start of an evaluation batch")
        crop = self.crop_batch(images, rois)
        preds = self._net(crop["image"])
        return self.to_image_coordinates(preds, crop["image_transform"], self.input_resolution)

    @staticmethod
    def to_image_coordinates(preds, image_transform: Tensor, N: int) -> Batch:
        """Predictions in the crop's [-1, 1] coordinates -> image pixels (reference :199-206: unnormalize_batch, then the
        image_backtransform that FocusRoi stored = the inverse of the crop's point transform).  `image_transform`: [B, 2, 3] image pixels ->
        crop pixels.  Pinned to the reference by tests/golden/eval.npz (oracle/tools/gen_golden_eval.py)."""
        # image pixels -> crop pixels -> [-1, 1]: invert the whole chain for the predictions
        to_norm = position_normalization(N, N).to(image_transform.device) @ Affine2d(image_transform)
        back = to_norm.inv()
        cats = {"coord": FieldCategory.xys, "pose": FieldCategory.quat, "pt3d_68": FieldCategory.points, "roi": FieldCategory.roi}
        out = {}
        for k, c in cats.items():
            if k in preds:
                v = preds[k]
                out[k] = apply_affine2d(back, k, v.value if hasattr(v, "value") else v, c)
        for k in ("coord_scales", "pose_scales_tril", "shapeparam", "unnormalized_quat"):
            if k in preds:
                out[k] = preds[k]
        meta = Metadata(N, int(image_transform.shape[0]), categories=dict(cats))
        return Batch(meta, out)

    def evaluate(self, metric, samples, batchsize: int = 128):
        """`samples`: iterable of dicts with "image", "roi" and the labels the metric reads (reference :210-222)."""
        for chunk in utils.iter_batched(samples, batchsize):
            images = [s["image"] for s in chunk]
            keys = [k for k in chunk[0] if k != "image"]
            targets = Batch(Metadata(0, len(chunk)), {k: torch.stack([torch.as_tensor(s[k]) for s in chunk]).to(self._device) for k in keys})
            targets["image_hw"] = torch.as_tensor([tuple(im.shape[:2]) for im in images])  # what AlignedRotationErrorMetric("perspective") reads
            preds = self.predict_batch(images, targets["roi"])
            metric.update(preds, targets)
        return metric.compute()


# ---------------------------------------------------------------------------------------------
# metrics (reference :295-440)
# ---------------------------------------------------------------------------------------------
class _SimpleConcatenatingErrorMetric:
    def __init__(self):
        self.error: list[Tensor] = []

    def update(self, preds: Batch, targets: Batch) -> None:
        self.error.append(self.compute_on_batch(preds, targets))

    def compute_on_batch(self, preds: Batch, targets: Batch) -> Tensor:
        raise NotImplementedError()

    def compute(self) -> Tensor:
        return torch.cat(self.error)

    def reset(self):
        self.error = []


class LabelExtractor(_SimpleConcatenatingErrorMetric):
    def __init__(self, key):
        super().__init__()
        self._key = key

    def compute_on_batch(self, preds, targets):
        return targets[self._key]


class PredExtractor(LabelExtractor):
    def compute_on_batch(self, preds, targets):
        return preds[self._key]


class GeodesicError(_SimpleConcatenatingErrorMetric):
    def compute_on_batch(self, preds, targets):
        return torchquaternion.geodesicdistance(targets["pose"], preds["pose"])


def _angle_errors(euler1, euler2):
    v1 = np.stack([np.cos(euler1), np.sin(euler1)], axis=-1)
    v2 = np.stack([np.cos(euler2), np.sin(euler2)], axis=-1)
    return np.arccos(np.clip(np.sum(v1 * v2, axis=-1), -1.0, 1.0))


def _quat_to_aflw3d_rotations(quats: Tensor):
    return np.array([utils.inv_aflw_rotation_conversion(q) for q in utils.convert_to_rot(quats.detach().cpu().numpy())])


def _aflw3d_euler_errors(quats1: Tensor, quats2: Tensor) -> Tensor:
    return torch.from_numpy(_angle_errors(_quat_to_aflw3d_rotations(quats1), _quat_to_aflw3d_rotations(quats2))).to(quats1.device)


class EulerAngleErrors(_SimpleConcatenatingErrorMetric):
    """|pitch|, |yaw|, |roll| differences in the AFLW2000-3D convention, radians, [B,3]."""

    def compute_on_batch(self, preds, targets):
        return _aflw3d_euler_errors(preds["pose"], targets["pose"])


class NormalizedXYSError(_SimpleConcatenatingErrorMetric):
    def compute_on_batch(self, preds, targets):
        x0, y0, x1, y1 = targets["roi"].unbind(-1)
        return torch.abs(preds["coord"] - targets["coord"]) / (x1 - x0)[:, None]


def _eval_keypoints(pred: Tensor, gt: Tensor, dims=3):
    assert pred.shape == gt.shape and pred.shape[-1] == 3
    pred, gt = pred.clone(), gt.clone()
    pred[:, :, 2] -= torch.mean(pred[:, :, 2], dim=-1, keepdim=True)
    gt[:, :, 2] -= torch.mean(gt[:, :, 2], dim=-1, keepdim=True)
    dist = torch.mean(torch.norm(pred[:, :, :dims] - gt[:, :, :dims], dim=-1), dim=-1)
    bbox = torch.sqrt((gt[:, :, 0].amax(1) - gt[:, :, 0].amin(1)) * (gt[:, :, 1].amax(1) - gt[:, :, 1].amin(1)))
    return dist / bbox


class UnweightedKptNME(_SimpleConcatenatingErrorMetric):
    def __init__(self, dimensions=3):
        super().__init__()
        self.dims = dimensions

    def compute_on_batch(self, preds, targets):
        return _eval_keypoints(preds["pt3d_68"], targets["pt3d_68"], self.dims)


class KptNmeResults(NamedTuple):
    bin_30_nme: float
    bin_60_nme: float
    bin_90_nme: float
    avg_nme: float


class KptNME:
    """Landmark NME binned by |yaw| of the label: 0-30, 30-60, 60-90 degrees (reference :407-440)."""

    def __init__(self, dimensions=3):
        self.dims, self.error, self.masks = dimensions, [], []

    def update(self, preds, targets):
        pyr = _quat_to_aflw3d_rotations(targets["pose"])
        yaw = np.abs(pyr[:, 1]) * 180.0 / np.pi
        self.masks.append(torch.from_numpy(np.stack([(a <= yaw) & (yaw < b) for a, b in ((0.0, 30.0), (30.0, 60.0), (60.0, 90.0))], -1)))
        self.error.append(_eval_keypoints(preds["pt3d_68"], targets["pt3d_68"], self.dims).cpu())

    def compute(self) -> KptNmeResults:
        errors, masks = torch.cat(self.error), torch.cat(self.masks).unbind(-1)
        bins = [torch.mean(errors[m]).item() for m in masks]
        return KptNmeResults(*bins, float(np.average(bins)))


def pose_error_table(euler_errors: Tensor, geodesic: Tensor) -> dict:
    """The summary row scripts/evaluate_pose_network.py prints: MAE per angle and their mean in degrees (reference
    scripts/evaluate_pose_network.py:205-291)."""
    e = euler_errors.detach().cpu().numpy() * utils.rad2deg
    mae = e.mean(0)
    return {"pitch": float(mae[0]), "yaw": float(mae[1]), "roll": float(mae[2]), "mae": float(mae.mean()),
            "geodesic": float(geodesic.detach().cpu().numpy().mean() * utils.rad2deg)}


# ---------------------------------------------------------------------------------------------
# rotation errors after an alignment of the predictions (reference :443-600; scripts/evaluate_pose_network.py --alignment-scheme)
# ---------------------------------------------------------------------------------------------
def compute_mean_rotation(rots, tol=0.0001, max_iter=100000):
    """Karcher mean of scipy Rotations inside the ball of radius pi/2 (the OPAL paper's evaluator, reference :447-459): start at the first
    one, move along the mean tangent vector until it is shorter than `tol`."""
    from scipy.spatial.transform import Rotation

    rots = rots[rots.magnitude() < np.pi / 2]
    mean = rots[0]
    for _ in range(max_iter):
        step = np.mean((mean.inv() * rots).as_rotvec(), axis=0)
        if np.linalg.norm(step) < tol:
            break
        mean = mean * Rotation.from_rotvec(step)
    return mean


def compute_opal_paper_alignment(pose_pred: Tensor, pose_target: Tensor, cluster_ids) -> Tensor:
    """Predictions with the mean offset to the labels removed, separately for every cluster id (the recorded individuals of Biwi):
    P <- P * mean(T^-1 P)^-1 (reference :462-482).  CPU tensors."""
    from scipy.spatial.transform import Rotation

    assert pose_pred.device.type == "cpu" and pose_target.device.type == "cpu"
    cluster_ids = np.asarray(cluster_ids)
    out = torch.empty_like(pose_pred)
    for cid in np.unique(cluster_ids):
        mask = cluster_ids == cid
        pred, target = Rotation.from_quat(pose_pred[mask].numpy()), Rotation.from_quat(pose_target[mask].numpy())
        offset = compute_mean_rotation(target.inv() * pred)
        out[torch.from_numpy(mask)] = torch.from_numpy((pred * offset.inv()).as_quat()).to(pose_pred.dtype)
    return out


class PerspectiveCorrector:
    """The network sees a face off the optical axis under the angle of its viewing ray; this rotates the predicted pose from the ray's frame
    into the camera frame (reference :485-544).  `fov`: horizontal field of view in degrees."""

    def __init__(self, fov):
        self._fov = fov
        self.f = 1.0 / np.tan(fov * np.pi / 180.0 * 0.5)

    def corrected_rotation(self, image_sizes: Tensor, coord: Tensor, pose: Tensor) -> Tensor:
        """image_sizes [B, 2] = (width, height); coord [B, 3] in pixels; pose [B, 4]."""
        half = 0.5 * image_sizes
        xy = (coord[..., :2] - half) / half[0]  # (the reference divides by the FIRST row's half size: all images of a set share one size)
        ray = torch.cat([xy, torch.as_tensor(self.f, device=xy.device, dtype=xy.dtype).expand_as(xy[..., :1])], dim=-1)
        return torchquaternion.mult(torchquaternion.from_matrix(self._make_look_at_matrix(ray)), pose)

    @staticmethod
    def _make_look_at_matrix(pos: Tensor) -> Tensor:
        """Columns x, y, z with z along `pos` and x kept horizontal."""
        z = pos / torch.norm(pos, dim=-1, keepdim=True)
        x = torch.cross(*torch.broadcast_tensors(pos.new_tensor([0.0, 1.0, 0.0]), z), dim=-1)
        x = x / torch.norm(x, dim=-1, keepdim=True)
        y = torch.cross(z, x, dim=-1)
        y = y / torch.norm(x, dim=-1, keepdim=True)  # (x is already unit length: the reference's normalisation of y is a no-op)
        return torch.stack([x, y, z], dim=-1)


class AlignedRotationErrorMetric:
    """Euler-angle or geodesic errors after `correction_mode` "perspective" (needs the image sizes: targets["image_hw"] [B, 2] as
    Predictor.evaluate provides it, or targets["image"] as a list of [H, W(, C)] frames) or "opal23" (needs targets["individual"]) -
    reference :547-600."""

    def __init__(self, error_mode, correction_mode, fov=None):
        assert error_mode in ("euler", "geo") and correction_mode in ("perspective", "opal23")
        self._error_mode, self._correction_mode, self._fov = error_mode, correction_mode, fov
        self.reset()

    def reset(self):
        self.image_sizes, self.target_quats, self.pred_quats, self.pred_coord, self.individual = [], [], [], [], []

    def update(self, preds: Batch, targets: Batch) -> None:
        self.target_quats.append(targets["pose"].cpu())
        self.pred_quats.append(preds["pose"].cpu())
        self.pred_coord.append(preds["coord"].cpu())
        if self._correction_mode == "perspective":
            hw = targets["image_hw"] if "image_hw" in targets else torch.as_tensor([tuple(t.shape[:2]) for t in targets["image"]])
            self.image_sizes.append(torch.as_tensor(hw).cpu())  # (h, w)
        else:
            self.individual.append(torch.as_tensor(targets["individual"]).cpu())

    def compute(self) -> Tensor:
        target, pred, coord = torch.cat(self.target_quats), torch.cat(self.pred_quats), torch.cat(self.pred_coord)
        if self._correction_mode == "perspective":
            wh = torch.flip(torch.cat(self.image_sizes), dims=(-1,))
            pred = PerspectiveCorrector(self._fov).corrected_rotation(wh, coord, pred)
        else:
            pred = compute_opal_paper_alignment(pred, target, torch.cat(self.individual).numpy())
        if self._error_mode == "euler":
            return _aflw3d_euler_errors(pred, target)
        return torchquaternion.geodesicdistance(pred, target)
