"""The evaluation table (the part around the metrics) against tests/golden/eval_table.npz, produced by the reference's
eval.py and scripts/evaluate_pose_network.py (oracle/tools/gen_golden_eval_table.py): alignment schemes, box-configuration names, the row
arithmetic of report() and the rendered tables.  CPU; tests/test_eval_table_gpu.py runs the script on the MI355X."""
import importlib.util
import json
import os
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
G = np.load(os.path.join(HERE, "golden", "eval_table.npz"))
T = lambda k: torch.from_numpy(G[k].copy())


def _script():
    spec = importlib.util.spec_from_file_location("amd_eval_script", os.path.join(REPO, "neuralnet-tracker-traincode_amd", "scripts", "evaluate_pose_network.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_alignment_schemes_match_the_reference():
    from scipy.spatial.transform import Rotation
    from trackertraincode import eval as E

    mean = E.compute_mean_rotation(Rotation.from_quat(G["al_pose_target"]).inv() * Rotation.from_quat(G["al_pose_pred"]))
    np.testing.assert_allclose(mean.as_quat(), G["mean_rotation_quat"], atol=1e-12)
    al = E.compute_opal_paper_alignment(T("al_pose_pred"), T("al_pose_target"), G["al_individual"])
    np.testing.assert_allclose(al.numpy(), G["opal_aligned"], atol=1e-7)
    wh = torch.flip(T("al_image_hw"), dims=(-1,))
    pc = E.PerspectiveCorrector(float(G["al_fov"])).corrected_rotation(wh, T("al_coord_pred"), T("al_pose_pred"))
    np.testing.assert_allclose(pc.numpy(), G["perspective_corrected"], atol=3e-7)
    for mode in ("opal23", "perspective"):
        for em in ("euler", "geo"):
            m = E.AlignedRotationErrorMetric(em, mode, float(G["al_fov"]))
            for lo in (0, 13, 29):  # ragged updates
                sl = slice(lo, {0: 13, 13: 29, 29: 40}[lo])
                m.update({"pose": T("al_pose_pred")[sl], "coord": T("al_coord_pred")[sl]},
                         {"pose": T("al_pose_target")[sl], "image_hw": T("al_image_hw")[sl], "individual": T("al_individual")[sl]})
            np.testing.assert_allclose(m.compute().numpy(), G[f"aligned_{em}_{mode}"], atol=3e-7, err_msg=f"{mode} {em}")
    # the correction must matter on this input, or the comparison above pins nothing
    assert np.abs(G["opal_aligned"] - G["al_pose_pred"]).max() > 1e-2 and np.abs(G["perspective_corrected"] - G["al_pose_pred"]).max() > 1e-2


def test_roi_config_names_and_table_match_the_reference(monkeypatch):
    S = _script()
    names = json.loads(str(G["roi_config_names"]))
    assert [str(c) for c in S.comprehensive_roi_configs] + [str(S.RoiConfig()), str(S.RoiConfig(1.3, True))] == names
    assert [list(c) for c in S.comprehensive_roi_configs] == json.loads(str(G["roi_config_fields"]))
    res = {k[len("res_"):]: G[k] for k in G.files if k.startswith("res_")}
    for with_points in (True, False):
        calls = [0]

        def fake_evaluate(fn, ds, cfg, args, _p=with_points):
            f = np.float32(1.0 + 0.125 * calls[0])  # the generator's rule: results * (1 + call / 8)
            calls[0] += 1
            r = {k: torch.from_numpy(res[k] * f) for k in ("pose_errs", "geodesic_errs", "euler_errs")}
            if _p:
                from trackertraincode.eval import KptNmeResults
                r["uw_nme_3d"] = torch.from_numpy(res["uw_nme_3d"] * f)
                r["nme_2d"] = KptNmeResults(*[float(x * f) for x in res["nme_2d"].tolist()])  # python float * float32, as in the generator
            return r

        monkeypatch.setattr(S, "evaluate", fake_evaluate)
        tb = S.TableBuilder()
        args = types.SimpleNamespace(alignment_scheme="none", device="cpu", vis="none", datadir=None)
        for fn in ("/models/run1/best.ckpt", "/models/run2/best.ckpt"):
            for ds in ("aflw2k3d", "biwi"):
                for cfg in (S.RoiConfig(), S.RoiConfig(1.2, False, False)):
                    S.report(fn, ds, cfg, args, tb)
        assert tb.build() == str(G[f"table_points{int(with_points)}"])
        ref_json = str(G[f"json_points{int(with_points)}"])
        mine = json.loads(tb.build_json())
        if ref_json.startswith("TypeError"):  # the reference's --json raises with landmark columns (numpy float32 in json.dumps); here it works
            assert set(mine) == {"../run1/best.ckpt", "../run2/best.ckpt"} and len(mine["../run1/best.ckpt"]["NME3d%"]) == 4  # (commonprefix is character-wise)
        else:
            ref = json.loads(ref_json)
            assert set(mine) == set(ref)
            for model in ref:
                assert list(mine[model]) == list(ref[model])
                for col in ref[model]:
                    a, b = mine[model][col], ref[model][col]
                    if isinstance(b[0], str):
                        assert a == b
                    else:
                        np.testing.assert_allclose(a, b, rtol=1e-6)


def test_validation_dataset_from_shards(tmp_path):
    """make_validation_dataset over the converted mini AFLW2000-3D file: extreme-pose filter, landmark boxes, single samples."""
    import shutil

    import trackertraincode.pipelines as P
    from trackertraincode import eval as E

    shutil.copy(os.path.join(HERE, "golden", "aflw2kmini.npz"), tmp_path / "aflw2k.npz")
    ds = P.make_validation_loader("aflw2k3d", use_head_roi=False, datadir=str(tmp_path))
    raw = np.load(tmp_path / "aflw2k.npz")
    pyr = E._quat_to_aflw3d_rotations(torch.from_numpy(raw["quats"]))  # pinned to the reference by tests/golden/eval.npz
    keep = np.nonzero((np.abs(pyr) < np.pi * 99 / 180).all(1) & (raw["coords"][:, 2] >= 0))[0]
    assert len(ds) == len(keep)
    samples = list(ds)
    assert [int(s["index"]) for s in samples] == keep.tolist()
    s = samples[0]
    assert s["image"].dtype == torch.uint8 and s["image"].dim() == 2
    xy = s["pt3d_68"][:, :2]
    assert torch.equal(s["roi"], torch.cat([xy.amin(0), xy.amax(0)]))  # (F_roi): PutRoiFromLandmarks(extend_to_forehead=False)
    np.testing.assert_allclose(s["coord"][:2].numpy(), raw["coords"][keep[0], :2] + 0.5)  # offset_points_by_half_pixel
    sub = P.make_validation_dataset("aflw2k3d", order=[2, 0], use_head_roi=False, datadir=str(tmp_path))
    assert [int(x["index"]) for x in sub] == [keep[2], keep[0]]
    # one extreme pose and one negative size drop out
    q = raw["quats"].copy()
    q[1] = [0.0, np.sin(np.deg2rad(120) / 2), 0.0, np.cos(np.deg2rad(120) / 2)]  # yaw 120 degrees
    c = raw["coords"].copy()
    c[2, 2] = -1.0
    idx = P.indices_without_extreme_poses(q, c)
    assert 1 not in idx and 2 not in idx and set(idx) <= set(range(len(q)))
    with pytest.raises(ValueError):
        P.make_validation_loader("nosuchset", datadir=str(tmp_path))
    with pytest.raises(FileNotFoundError):
        P.make_validation_loader("biwi", use_head_roi=False, datadir=str(tmp_path))
    have_blob = os.path.exists(os.path.join(os.path.dirname(P.__file__), "facemodel", "bfm_noneck_v3.pkl"))
    if not have_blob:
        with pytest.raises(FileNotFoundError, match="bfm_noneck_v3"):
            P.make_validation_loader("aflw2k3d", datadir=str(tmp_path))  # the reference's default (H_roi) needs the head mesh


def test_individuals_from_sequence_starts(tmp_path):
    from trackertraincode.datasets.shards import decode_pose_shard

    raw = dict(np.load(os.path.join(HERE, "golden", "aflw2kmini.npz")))
    n = len(raw["rois"])
    raw["sequence_starts"] = np.array([0, 5, 6, n])
    np.savez(tmp_path / "seq.npz", **raw)
    ind = decode_pose_shard(str(tmp_path / "seq.npz"))["individual"]
    assert ind.tolist() == [0] * 5 + [1] + [2] * (n - 6)
