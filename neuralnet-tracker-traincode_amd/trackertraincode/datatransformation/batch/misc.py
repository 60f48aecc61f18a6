"""`PutRoiFromLandmarks` (reference: datatransformation/batch/misc.py:9-31): replaces a sample's face box by the xy extent of its 68
landmarks or - `extend_to_forehead=True` - of every vertex of the posed BFM head mesh (the box then reaches over the forehead; the
evaluation script's `(H_roi)` boxes and `roi_override="extent_to_forehead"`).  Works on one sample or on a whole batch (leading
dimensions are kept), on whichever device the labels live.  The mesh variant needs the BFM blob (facemodel/bfm.py).  Like the reference
it poses the MEAN head: its test for shape parameters looks up the key "shapeparams", which no sample carries."""
from __future__ import annotations

import torch

from ...facemodel.bfm import BFMModel, ScaledBfmModule
from ...neuralnets.modelcomponents import rigid_transformation_25d
from ...neuralnets.rotrepr import QuatRepr


def head_extent_roi(vertices: torch.Tensor, coord: torch.Tensor, pose: torch.Tensor, chunk: int = 256) -> torch.Tensor:
    """[..., 4] = (min x, min y, max x, max y) of the mesh `vertices` [V, 3] posed by coord [..., 3] (x, y, size) and pose [..., 4]."""
    lead = coord.shape[:-1]
    c, q = coord.reshape(-1, 3), pose.reshape(-1, 4)
    out = torch.empty((c.shape[0], 4), dtype=torch.float32, device=c.device)
    for i in range(0, c.shape[0], chunk):  # [chunk, V, 3] at a time: the mesh has tens of thousands of vertices
        pts = rigid_transformation_25d(QuatRepr(q[i:i + chunk]), c[i:i + chunk, :2], c[i:i + chunk, 2:], vertices)
        out[i:i + chunk, :2] = pts[..., :2].amin(dim=-2)
        out[i:i + chunk, 2:] = pts[..., :2].amax(dim=-2)
    return out.reshape(*lead, 4)


class PutRoiFromLandmarks:
    def __init__(self, extend_to_forehead=False, headmodel: ScaledBfmModule | None = None):
        self.extend_to_forehead = extend_to_forehead
        self.headmodel = headmodel
        if extend_to_forehead and headmodel is None:
            self.headmodel = ScaledBfmModule(BFMModel())  # FileNotFoundError when the blob is absent

    def _create_roi(self, landmarks3d: torch.Tensor, sample):
        if self.extend_to_forehead:
            verts = self.headmodel.vertices.to(landmarks3d.device)
            return head_extent_roi(verts, sample["coord"], sample["pose"])
        xy = landmarks3d[..., :2]
        return torch.cat([xy.amin(dim=-2), xy.amax(dim=-2)], dim=-1).to(torch.float32)

    def __call__(self, sample):
        if "pt3d_68" in sample:
            sample["roi"] = self._create_roi(sample["pt3d_68"], sample)
        return sample
