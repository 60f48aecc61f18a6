"""Converted pose-dataset shards -> frames resident in HBM.

The reference reads its training sets from HDF5 files through h5py in DataLoader worker processes and decodes one JPEG per sample
with OpenCV (datasets/dshdf5.py:59-336, dshdf5pose.py:198-256; file list pipelines.py:120-300).  This image has neither h5py nor
OpenCV in its default interpreter, and the MI355X path wants the frames on the GPU anyway (datasets/resident.py), so the job is
split:

  oracle/tools/h5_to_npz.py   (build container, /opt/conda python with h5py) one .h5 -> one .npz shard: the JPEG blobs untouched
                              (`image_bytes` + `image_lengths`, or raw `images`) and the label arrays under their HDF5 names
  decode_pose_shard()         here: PIL decodes every blob to grey uint8 ONCE, frames are zero-padded to a common size (labels
                              are pixel coordinates: padding right / bottom does not move them), names are mapped as the reference
                              maps them (dshdf5pose.py:34-46: rois -> roi, coords -> coord, quats -> pose, shapeparams -> shapeparam),
                              and the half-pixel offset of cell-centred pixels (batch/normalization.py:83-90: + 0.5 on the xy of
                              `coord` and `pt3d_68`) is applied
  load_resident_frames()      the decoded shard as a ResidentFrames on the device, with its task Tag
"""
from __future__ import annotations

import io

import numpy as np
import torch

from .resident import ResidentFrames

# dshdf5pose.py:33-46 (its whitelist :168-180 also names semseg / seg_image, which no pose dataset of the training script carries)
_NAME_MAP = {"rois": "roi", "coords": "coord", "quats": "pose", "pt3d_68": "pt3d_68", "pt2d_68": "pt2d_68", "shapeparams": "shapeparam", "hasface": "hasface"}


def _to_grey(im: np.ndarray) -> np.ndarray:
    """Raw frames with a colour axis -> grey with the luma weights the JPEG path uses (PIL "L" = ITU-R 601: 299/587/114 per mille,
    what cv2.imdecode(..., IMREAD_GRAYSCALE) of the reference computes too), rounded to nearest."""
    if im.ndim == 2:
        return np.asarray(im, dtype=np.uint8)
    rgb = np.asarray(im[..., :3], dtype=np.float32)
    return np.clip(np.rint(rgb @ np.array([0.299, 0.587, 0.114], np.float32)), 0, 255).astype(np.uint8)


def _decode_grey(blob: bytes) -> np.ndarray:
    from PIL import Image  # (Pillow is in the image; OpenCV is not)

    return np.asarray(Image.open(io.BytesIO(blob)).convert("L"), dtype=np.uint8)


def decode_pose_shard(path: str, half_pixel_offset: bool = True) -> dict:
    """{"image": uint8 [N,1,H,W] (zero-padded to the largest frame), "image_size": int32 [N,2] (w, h), labels...}"""
    d = np.load(path)
    if "image_bytes" in d.files:
        lengths = d["image_lengths"].astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(lengths)])
        blob = d["image_bytes"]
        frames = [_decode_grey(blob[offs[i]:offs[i + 1]].tobytes()) for i in range(len(lengths))]
    elif "images" in d.files:
        imgs = d["images"]
        frames = [_to_grey(im) for im in imgs]
    else:
        raise ValueError(f"{path}: neither image_bytes/image_lengths nor images")
    H, W = max(f.shape[0] for f in frames), max(f.shape[1] for f in frames)
    image = np.zeros((len(frames), 1, H, W), np.uint8)
    for i, f in enumerate(frames):
        image[i, 0, :f.shape[0], :f.shape[1]] = f
    out = {"image": image, "image_size": np.array([[f.shape[1], f.shape[0]] for f in frames], np.int32)}
    for src, dst in _NAME_MAP.items():
        if src in d.files:
            v = np.asarray(d[src])
            out[dst] = v.astype(np.float32) if v.dtype.kind == "f" else v
    # who is in a frame (reference dshdf5pose.py:221-236): from the sequence boundaries, else the stored array
    if "sequence_starts" in d.files:
        starts = np.asarray(d["sequence_starts"]).astype(np.int32)
        out["individual"] = np.concatenate([np.full(b - a, i, dtype=np.int32) for i, (a, b) in enumerate(zip(starts[:-1], starts[1:]))])
    elif "individual" in d.files:
        out["individual"] = np.asarray(d["individual"]).astype(np.int32)
    if half_pixel_offset:
        if "coord" in out:
            out["coord"] = out["coord"].copy()
            out["coord"][:, :2] += 0.5
        for k in ("pt3d_68", "pt2d_68"):
            if k in out:
                out[k] = out[k].copy()
                out[k][..., :2] += 0.5
    n = {len(v) for v in out.values()}
    if len(n) != 1:
        raise ValueError(f"{path}: fields of different lengths")
    return out


def load_resident_frames(path: str, tag, device="cuda", coord_convention_id: int = 0, half_pixel_offset: bool = True) -> ResidentFrames:
    shard = decode_pose_shard(path, half_pixel_offset)
    fields = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in shard.items() if k != "image_size"}
    fields["coord_convention_id"] = torch.full((len(shard["image"]),), int(coord_convention_id), dtype=torch.int32, device=device)
    return ResidentFrames(tag, fields)
