#!/usr/bin/env python3
"""Golden vectors for the full-mesh head model and the forehead box, produced by the REFERENCE's own classes (build container only):

  facemodel/bfm.py:23-97                  BFMModel + ScaledBfmModule, run on a seeded SYNTHETIC blob of the missing files' format
                                          (oracle/synth.py write_synthetic_bfm_blob; the reference reads it through a patched folder
                                          variable - its real bfm_noneck_v3.pkl / tri.pkl are .MISSING_LARGE_BLOBS)
  datatransformation/batch/misc.py:9-31   PutRoiFromLandmarks(extend_to_forehead=True / False) on single samples
  neuralnets/modelcomponents.py:38-56,85-94  rigid_transformation_25d / PosedDeformableHead underneath

-> tests/golden/bfm_head_roi.npz  (digests of the scaled arrays, full keypoints / tri, the boxes; inputs stored)
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shims  # noqa: E402

torch = ref_shims.install(synthetic_bfm=False)
import trackertraincode.facemodel.bfm as RB  # noqa: E402
from oracle.synth import digest, write_synthetic_bfm_blob  # noqa: E402

tmp = tempfile.mkdtemp()
write_synthetic_bfm_blob(tmp)
RB._current_folder = tmp  # the reference looks for its blobs beside bfm.py
from trackertraincode.datatransformation.batch.misc import PutRoiFromLandmarks  # noqa: E402  (imports pipelines -> Batch only)

full = RB.BFMModel()
out = {"blob_seed": np.array(515), "vertexcount": np.array(full.vertexcount)}
out["scaled_vertices"] = digest(full.scaled_vertices)
out["scaled_bases"] = digest(full.scaled_bases)
out["w_norm"] = full.w_norm
out["keypoints"] = full.keypoints
out["scaled_tri"] = full.scaled_tri
mod = RB.ScaledBfmModule(full)
rng = np.random.default_rng(99)
sp = (rng.standard_normal((3, 50)) * 0.5).astype(np.float32)
out["shapeparams"] = sp
out["deformed"] = np.stack([digest(mod(torch.from_numpy(sp[i])).numpy()) for i in range(3)])

n = 12
q = rng.standard_normal((n, 4)).astype(np.float32)
q /= np.linalg.norm(q, axis=-1, keepdims=True)
coord = np.concatenate([rng.uniform(100, 300, (n, 2)), rng.uniform(40, 90, (n, 1))], -1).astype(np.float32)
pts = (rng.standard_normal((n, 68, 3)) * 30 + np.array([200.0, 200.0, 0.0])).astype(np.float32)
out.update(pose=q, coord=coord, pt3d_68=pts)
head, face = PutRoiFromLandmarks(extend_to_forehead=True), PutRoiFromLandmarks(extend_to_forehead=False)
rh, rf = [], []
for i in range(n):
    s = {"pose": torch.from_numpy(q[i]), "coord": torch.from_numpy(coord[i]), "pt3d_68": torch.from_numpy(pts[i]),
         "shapeparam": torch.from_numpy(sp[i % 3])}  # present, and ignored: the reference tests for the key "shapeparams"
    rh.append(head(dict(s))["roi"].numpy())
    rf.append(face(dict(s))["roi"].numpy())
out["roi_head"], out["roi_face"] = np.stack(rh), np.stack(rf)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "bfm_head_roi.npz"), **out)
print("bfm_head_roi.npz:", {k: np.asarray(v).shape for k, v in out.items()})
