"""Batched crop augmentation on the MI355X: random view ROI -> affine warp to 129x129 -> label bookkeeping ->
normalisation, for a whole sub-batch per launch (csrc/warp.hip).  Replaces the reference's per-sample CPU
path (GeneralFocusRoi.__call__ + normalize_batch + whiten, datatransformation/batch/geometric.py:193-231,
batch/normalization.py:20-56, pipelines.py:508-532) for training from decoded frames resident in HBM."""
from __future__ import annotations

import torch

from .. import _hip
from ..datasets.batch import Batch, Metadata
from ..neuralnets.affine2d import Affine2d
from .batch.geometric import MakeRoiRandomizationParameters, RoiFocusRandomizationParameters

_p = _hip.ptr


class GpuFocusRoiAugment:
    """batch fields used: image u8/f32 [B,1,Hs,Ws] (grey levels 0..255), roi [B,4] (pixels), and - if present -
    coord [B,3], pose [B,4], pt3d_68 [B,68,3].  Returns a new Batch with image f32 [B,1,N,N] = crop/256 - 0.5
    and labels in the crop's [-1,1] coordinates; other fields pass through."""

    def __init__(self, new_size=129, rotation_aug_angle=30.0, extension_factor=1.1, beyond_border_shift=0.3, whiten=True,
                 make_params=None, flip_rot_p: float | None = None, roi_from_landmarks: bool = False):
        """flip_rot_p: the reference's `horizontal_flip_and_rot_90(p_rot)` behind the crop (pipelines.py:373-377, batch/geometric.py:234-267):
        every sample is mirrored with probability 1/2 and turned by +-90 degrees with probability p_rot / 2 each; None = off (the eval
        stage).  roi_from_landmarks: `roi_override="landmarks"` (pipelines.py:343-350, batch/misc.py:9-31): the face box the crop is taken
        around AND the box label of the crop are the xy extent of pt3d_68 (samples without landmarks keep their stored box)."""
        self.new_size = int(new_size)
        self.beyond_border_shift = float(beyond_border_shift)
        self.make_params = make_params or MakeRoiRandomizationParameters(rotation_aug_angle, extension_factor)
        self.mul, self.add = 1.0 / 256.0, (-0.5 if whiten else 0.0)
        self.flip_rot_p = None if flip_rot_p is None else float(flip_rot_p)
        self.roi_from_landmarks = bool(roi_from_landmarks)
        self._fliprot = None  # [6, 3, 3]: the six point transforms (rot_dir + 1) * 2 + do_flip in crop pixels, built on first use

    def fliprot_table(self) -> torch.Tensor:
        """The point transform of every (rot_dir, do_flip) pair exactly as the reference composes it (batch/geometric.py:241-251), for the
        N x N crop: row (rot_dir + 1) * 2 + do_flip."""
        if self._fliprot is None:
            import math

            N = float(self.new_size)
            rows = []
            for rot_dir in (-1, 0, 1):
                for do_flip in (0, 1):
                    tr = Affine2d.identity()
                    if rot_dir != 0:
                        tr = (tr @ Affine2d.range_remap_2d([-1.0, -1.0], [1.0, 1.0], [0.0, 0.0], [N, N])
                              @ Affine2d.trs(angles=torch.tensor(rot_dir * math.pi * 0.5, dtype=torch.float32))
                              @ Affine2d.range_remap_2d([0.0, 0.0], [N, N], [-1.0, -1.0], [1.0, 1.0]))
                    if do_flip:
                        tr = tr @ Affine2d.range_remap_2d([0.0, 0.0], [N, N], [N, 0], [0, N])
                    rows.append(tr.tensor33())
            self._fliprot = torch.stack(rows).to(torch.float32)
        return self._fliprot

    def draw_fliprot(self, B: int, generator: torch.Generator | None = None) -> torch.Tensor:
        """Codes (rot_dir + 1) * 2 + do_flip of B samples with the reference's probabilities (np.random.randint(0, 2) == 0;
        np.random.choice([-1, 0, 1], p=[p / 2, 1 - p, p / 2]))."""
        dev = generator.device if generator is not None else "cpu"
        u = torch.rand((B, 2), generator=generator, device=dev)
        flip = (u[:, 0] < 0.5).long()
        p = self.flip_rot_p
        rot = torch.where(u[:, 1] < p / 2, 0, torch.where(u[:, 1] < 1.0 - p / 2, 1, 2))  # rot_dir + 1
        return rot * 2 + flip

    def __call__(self, batch: Batch, generator: torch.Generator | None = None, params: RoiFocusRandomizationParameters | None = None,
                 fliprot_codes: torch.Tensor | None = None) -> Batch:
        img = batch["image"]
        if not img.is_cuda:
            raise RuntimeError("GpuFocusRoiAugment runs in HIP kernels: CUDA tensors required (no CPU fallback)")
        B, _, Hs, Ws = img.shape
        dev = img.device
        _hip.lib().clear_stale_error("GpuFocusRoiAugment")  # once per loader batch: these launches run before training_step's own clear
        if params is None:
            params = self.make_params((B,), generator=generator, device=dev if generator is None or generator.device.type == "cuda" else "cpu")
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        roi = f32(batch["roi"])
        if self.roi_from_landmarks and "pt3d_68" in batch:  # PutRoiFromLandmarks(extend_to_forehead=False) in front of the crop
            xy = f32(batch["pt3d_68"])[..., :2]
            roi = torch.cat((xy.amin(dim=-2), xy.amax(dim=-2)), dim=-1).contiguous()
        view = torch.empty((B, 4), dtype=torch.int32, device=dev)
        tr = torch.empty((B, 2, 3), dtype=torch.float32, device=dev)
        out_img = torch.empty((B, 1, self.new_size, self.new_size), dtype=torch.float32, device=dev)
        L = _hip.lib()
        # named: a temporary's memory would be handed to the next allocation before the kernel has read it
        scales, translations, angles = f32(params.scales), f32(params.translations), f32(params.angles)
        L.call("ttk_view_roi", _p(roi), _p(scales), _p(translations), self.beyond_border_shift, B, _p(view))
        L.call("ttk_roi_transform", _p(view), _p(angles), B, self.new_size, _p(tr))
        if self.flip_rot_p is not None or fliprot_codes is not None:
            # the mirror / quarter turn of the finished crop is an exact pixel permutation = the same affine map composed onto the crop's
            # transform: ONE warp samples the source at the permuted pixel centres, and the labels see the composed transform as in the
            # reference (which applies `tr` of :241-251 to every non-image field)
            codes = fliprot_codes if fliprot_codes is not None else self.draw_fliprot(B, generator)
            F = self.fliprot_table().to(dev)[codes.to(dev)]
            tr33 = torch.cat((tr, torch.tensor([0.0, 0.0, 1.0], device=dev).expand(B, 1, 3)), dim=1)
            tr = torch.bmm(F, tr33)[:, :2, :].contiguous()
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
        if self.roi_from_landmarks and pts_out is not None:  # PutRoiFromLandmarks behind the crop: the box label = extent of the crop's landmarks
            xy = pts_out[..., :2]
            new_roi = torch.cat((xy.amin(dim=-2), xy.amax(dim=-2)), dim=-1).contiguous()
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
