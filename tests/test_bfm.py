"""Full-mesh head model (facemodel/bfm.py) and the forehead box (datatransformation/batch/misc.py) against tests/golden/bfm_head_roi.npz:
the reference's BFMModel / ScaledBfmModule / PutRoiFromLandmarks run on a seeded synthetic blob of the missing files' format
(oracle/tools/gen_golden_bfm.py).  CPU."""
import os

import numpy as np
import pytest
import torch

from oracle.synth import digest_close, write_synthetic_bfm_blob

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bfm_head_roi.npz"))


@pytest.fixture(scope="module")
def blob(tmp_path_factory):
    d = tmp_path_factory.mktemp("bfm")
    write_synthetic_bfm_blob(str(d), seed=int(G["blob_seed"]))
    return str(d)


def test_scaled_arrays_match_the_reference(blob):
    from trackertraincode.facemodel.bfm import BFMModel, ScaledBfmModule

    m = BFMModel(folder=blob)
    assert m.vertexcount == int(G["vertexcount"])
    for name in ("scaled_vertices", "scaled_bases"):
        ok, msg = digest_close(G[name], getattr(m, name), rtol=1e-6, atol=1e-9)
        assert ok, f"{name}: {msg}"
    assert m.scaled_bases.shape == (50, m.vertexcount, 3) and m.scaled_vertices.shape == (m.vertexcount, 3)
    np.testing.assert_allclose(m.w_norm, G["w_norm"], rtol=1e-6)
    assert np.array_equal(m.keypoints, G["keypoints"]) and np.array_equal(m.scaled_tri, G["scaled_tri"])
    mod = ScaledBfmModule(m)
    assert mod.num_eigvecs == 50 and set(dict(mod.named_buffers())) == {"vertices", "deform_base", "tri", "keypoints"}
    sp = torch.from_numpy(G["shapeparams"])
    for i in range(3):
        ok, msg = digest_close(G["deformed"][i], mod(sp[i]).numpy(), rtol=1e-5, atol=1e-7)
        assert ok, msg
    assert mod(sp).shape == (3, m.vertexcount, 3)  # batched


def test_forehead_and_landmark_boxes_match_the_reference(blob):
    from trackertraincode.datatransformation.batch import PutRoiFromLandmarks
    from trackertraincode.facemodel.bfm import BFMModel, ScaledBfmModule

    head = PutRoiFromLandmarks(extend_to_forehead=True, headmodel=ScaledBfmModule(BFMModel(folder=blob)))
    face = PutRoiFromLandmarks(extend_to_forehead=False)
    T = lambda k: torch.from_numpy(G[k].copy())
    whole = {"pose": T("pose"), "coord": T("coord"), "pt3d_68": T("pt3d_68")}
    np.testing.assert_allclose(head(dict(whole))["roi"].numpy(), G["roi_head"], rtol=1e-5, atol=2e-3)   # pixels; boxes span hundreds
    np.testing.assert_allclose(face(dict(whole))["roi"].numpy(), G["roi_face"], rtol=0, atol=0)
    one = {k: v[3] for k, v in whole.items()}  # a single sample, as the reference's dataset transform sees it
    np.testing.assert_allclose(head(dict(one))["roi"].numpy(), G["roi_head"][3], rtol=1e-5, atol=2e-3)
    assert "roi" not in head({"pose": whole["pose"], "coord": whole["coord"]})  # no landmarks: left alone (reference :28-31)


def test_missing_blob_is_loud(tmp_path):
    from trackertraincode.datatransformation.batch import PutRoiFromLandmarks
    from trackertraincode.facemodel.bfm import BFMModel

    with pytest.raises(FileNotFoundError, match="bfm_noneck_v3.pkl"):
        BFMModel(folder=str(tmp_path))
    if not os.path.exists(os.path.join(os.path.dirname(__import__("trackertraincode").__file__), "facemodel", "bfm_noneck_v3.pkl")):
        with pytest.raises(FileNotFoundError):
            PutRoiFromLandmarks(extend_to_forehead=True)
