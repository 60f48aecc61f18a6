"""The full Basel-face-model head mesh (reference: facemodel/bfm.py:23-97): mean vertices, the 40 shape + 10 expression deformation
vectors, triangles and the 68 keypoint vertex indices, in the reference's units (a head about one unit wide, y and z flipped to image
conventions, centred between the ears).

The two data files - `bfm_noneck_v3.pkl` (dict: `u` [3V,1], `w_shp` [3V,>=40], `w_exp` [3V,>=10], `keypoints` [204]) and `tri.pkl`
([3,F]) of 3DDFA_V2 - are large blobs the reference's checkout does not carry (.MISSING_LARGE_BLOBS); this build does not carry them
either.  Put them into trackertraincode/facemodel/ (or pass `folder=`) to use what needs the full mesh: `roi_override=
"extent_to_forehead"` and the `(H_roi)` boxes of the evaluation script.  Without them `BFMModel()` raises FileNotFoundError - nothing is
substituted.  The arithmetic below is pinned by tests/golden/bfm_head_roi.npz: the reference's own BFMModel run on a seeded synthetic
blob of the same format (oracle/tools/gen_golden_bfm.py)."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch
import torch.nn as nn

_FOLDER = os.path.dirname(os.path.abspath(__file__))
# vertex indices of the eye contours that replace the stored ones (reference :40-43: the stored positions do not survive the
# closed-eye deformations); keypoint slots in 68-point order
_LEFT_EYE = ((36, 37, 38, 39, 41, 40), (1959, 3887, 5048, 6216, 3513, 4674))
_RIGHT_EYE = ((42, 43, 44, 45, 47, 46), (9956, 11223, 12384, 14327, 11495, 12656))
_AXIS_SIGNS = np.array([1.0, -1.0, -1.0], dtype=np.float32)  # model axes -> image axes (y down, z into the screen)
_CENTER = np.array([0.0, -0.26, -0.9], dtype=np.float32)


def _unpickle(path):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not found: the BFM head mesh (bfm_noneck_v3.pkl + tri.pkl of 3DDFA_V2) is a large blob that neither the "
                                "reference's checkout nor this package carries; copy both files next to facemodel/bfm.py or pass folder=")
    with open(path, "rb") as f:
        return pickle.load(f)


class BFMModel:
    def __init__(self, shape_dim=40, exp_dim=10, folder: str | None = None):
        folder = folder or _FOLDER
        blob = _unpickle(os.path.join(folder, "bfm_noneck_v3.pkl"))
        self.u = np.asarray(blob["u"], dtype=np.float32)
        self.w_shp = np.asarray(blob["w_shp"], dtype=np.float32)[..., :shape_dim]
        self.w_exp = np.asarray(blob["w_exp"], dtype=np.float32)[..., :exp_dim]
        self.vertexcount = self.u.shape[0] // 3
        self.tri = np.ascontiguousarray(np.asarray(_unpickle(os.path.join(folder, "tri.pkl"))).T).astype(np.int32)  # [F, 3]
        kp = np.asarray(blob["keypoints"]).astype(np.int64)[::3] // 3  # stored per coordinate (3 * vertex + axis)
        for slots, verts in (_LEFT_EYE, _RIGHT_EYE):
            kp[list(slots)] = verts
        self.keypoints = kp
        self.w_norm = np.linalg.norm(np.concatenate((self.w_shp, self.w_exp), axis=1), axis=0)

    def _basis(self, w, scale):
        return (scale * w.reshape(self.vertexcount, 3, -1)).transpose(2, 0, 1) * _AXIS_SIGNS  # [n, V, 3]

    @property
    def scaled_shp_base(self):
        return self._basis(self.w_shp, 20.0)

    @property
    def scaled_exp_base(self):
        return self._basis(self.w_exp, 5.0e-5)

    @property
    def scaled_bases(self):
        """[40 + 10, V, 3]"""
        return np.concatenate([self.scaled_shp_base, self.scaled_exp_base], axis=0)

    @property
    def scaled_vertices(self):
        """[V, 3]"""
        return np.ascontiguousarray(self.u.reshape(-1, 3) * np.float32(1.0e-5) * _AXIS_SIGNS - _CENTER)

    @property
    def scaled_tri(self):
        return np.ascontiguousarray(self.tri[..., ::-1])  # winding order for the flipped axes


class ScaledBfmModule(nn.Module):
    """The mesh as a module: forward(shapeparams [..., 50]) -> vertices [..., V, 3] (reference :80-97)."""

    def __init__(self, original: BFMModel):
        super().__init__()
        self.register_buffer("vertices", torch.from_numpy(np.asarray(original.scaled_vertices, dtype=np.float32)))
        self.register_buffer("deform_base", torch.from_numpy(np.asarray(original.scaled_bases, dtype=np.float32)))
        self.register_buffer("tri", torch.from_numpy(np.asarray(original.scaled_tri)))
        self.register_buffer("keypoints", torch.from_numpy(np.asarray(original.keypoints)).to(torch.long))

    @property
    def num_eigvecs(self):
        return self.deform_base.size(0)

    def forward(self, shapeparams):
        return torch.einsum("...i,ivd->...vd", shapeparams, self.deform_base) + self.vertices
