"""Batched crop augmentation on the MI355X: random view ROI -> affine warp to 129x129 -> label bookkeeping ->
normalisation, for a whole sub-batch per launch (csrc/warp.hip).  Replaces the reference's per-sample CPU
path (GeneralFocusRoi.__call__ + normalize_batch + whiten, datatransformation/batch/geometric.py:193-231,
batch/normalization.py:20-56, pipelines.py:508-532) for training from decoded frames resident in HBM."""
from __future__ import annotations

import torch

from .. import _hip
from ..datasets.batch import Batch, Metadata
from .batch.geometric import MakeRoiRandomizationParameters, RoiFocusRandomizationParameters

_p = _hip.ptr


class GpuFocusRoiAugment:
    """batch fields used: image u8/f32 [B,1,Hs,Ws] (grey levels 0..255), roi [B,4] (pixels), and - if present -
    coord [B,3], pose [B,4], pt3d_68 [B,68,3].  Returns a new Batch with image f32 [B,1,N,N] = crop/256 - 0.5
    and labels in the crop's [-1,1] coordinates; other fields pass through."""

    def __init__(self, new_size=129, rotation_aug_angle=30.0, extension_factor=1.1, beyond_border_shift=0.3, whiten=True,
                 make_params=None):
        self.new_size = int(new_size)
        self.beyond_border_shift = float(beyond_border_shift)
        self.make_params = make_params or MakeRoiRandomizationParameters(rotation_aug_angle, extension_factor)
        self.mul, self.add = 1.0 / 256.0, (-0.5 if whiten else 0.0)

    def __call__(self, batch: Batch, generator: torch.Generator | None = None, params: RoiFocusRandomizationParameters | None = None) -> Batch:
        img = batch["image"]
        if not img.is_cuda:
            raise RuntimeError("GpuFocusRoiAugment runs in HIP kernels: CUDA tensors required (no CPU fallback)")
        B, _, Hs, Ws = img.shape
        dev = img.device
        if params is None:
            params = self.make_params((B,), generator=generator, device=dev if generator is None or generator.device.type == "cuda" else "cpu")
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        roi = f32(batch["roi"])
        view = torch.empty((B, 4), dtype=torch.int32, device=dev)
        tr = torch.empty((B, 2, 3), dtype=torch.float32, device=dev)
        out_img = torch.empty((B, 1, self.new_size, self.new_size), dtype=torch.float32, device=dev)
        L = _hip.lib()
        # named: a temporary's memory would be handed to the next allocation before the kernel has read it
        scales, translations, angles = f32(params.scales), f32(params.translations), f32(params.angles)
        L.call("ttk_view_roi", _p(roi), _p(scales), _p(translations), self.beyond_border_shift, B, _p(view))
        L.call("ttk_roi_transform", _p(view), _p(angles), B, self.new_size, _p(tr))
        src = img.contiguous()
        if src.dtype not in (torch.uint8, torch.float32):
            src = src.float()
        L.call("ttk_affine_warp", _p(src), int(src.dtype == torch.uint8), B, Hs, Ws, _p(tr), _p(out_img), self.new_size, self.mul, self.add)
        out = {k: v for k, v in batch.items()}
        out["image"] = out_img
        coord = f32(batch["coord"]).clone() if "coord" in batch else None
        pose = f32(batch["pose"]).clone() if "pose" in batch else None
        new_roi = roi.clone()
        pts_in = f32(batch["pt3d_68"]) if "pt3d_68" in batch else None
        pts_out = torch.empty_like(pts_in) if pts_in is not None else None
        L.call("ttk_affine_labels", _p(tr), B, self.new_size, _p(coord), _p(pose), _p(new_roi), _p(pts_in), _p(pts_out))
        out["roi"] = new_roi
        if coord is not None:
            out["coord"] = coord
        if pose is not None:
            out["pose"] = pose
        if pts_out is not None:
            out["pt3d_68"] = pts_out
        meta = Metadata(self.new_size, batch.meta.batchsize, batch.meta.tag, batch.meta.seq, dict(batch.meta.categories))
        res = Batch(meta, out)
        res.view_roi, res.transform = view, tr  # exposed for tests / back-transforms
        return res
