"""scripts/evaluate_pose_network.py end to end on the MI355X: checkpoint -> validation samples -> Predictor (HIP crop + network) -> metrics ->
table, against the same rows computed with the CPU oracle's network on the same crops."""
import importlib.util
import json
import os
import shutil

import numpy as np
import pytest
import torch

from util import GOLDEN, build_net, load_golden

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script():
    spec = importlib.util.spec_from_file_location("amd_eval_script", os.path.join(REPO, "neuralnet-tracker-traincode_amd", "scripts", "evaluate_pose_network.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_script_tables(tmp_path, capsys):
    from trackertraincode import eval as E
    from trackertraincode.neuralnets import models

    S = _script()
    d, meta = load_golden("model_default.npz")
    cal = {k[len("calib/"):]: d[k] for k in d.files if k.startswith("calib/")}
    net = build_net(meta, "cpu", cal)
    ck = str(tmp_path / "run1" / "best.ckpt")
    os.makedirs(os.path.dirname(ck))
    models.save_model(net, ck)
    shutil.copy(os.path.join(GOLDEN, "aflw2kmini.npz"), tmp_path / "aflw2k.npz")
    out_json = str(tmp_path / "t.json")
    S.main([ck, "--ds", "aflw2k3d", "--datadir", str(tmp_path), "--comprehensive-roi", "--json", out_json, "--allow-landmark-roi-fallback"])
    err = capsys.readouterr().err
    table = json.load(open(out_json))
    (model, cols), = table.items()
    have_blob = "note: no BFM head mesh" not in err
    assert len(cols["Data"]) == (6 if have_blob else 3)
    assert cols["Data"][-3:] == ["AFLW 2k 3d / (F_roi)ROI1.2", "AFLW 2k 3d / (F_roi)ROI1.1", "AFLW 2k 3d / (F_roi)ROI1.0"]
    # the same rows from the metrics driven by hand
    samples = list(S.pipelines.make_validation_loader("aflw2k3d", use_head_roi=False, datadir=str(tmp_path)))
    pred = E.Predictor(models.load_model(ck), 1.1, device="cuda")
    eul, geo, xys, nme = E.EulerAngleErrors(), E.GeodesicError(), E.NormalizedXYSError(), E.KptNME(dimensions=2)

    class All:
        def update(self, p, t):
            for m in (eul, geo, xys, nme):
                m.update(p, t)

        def compute(self):
            return None

    pred.evaluate(All(), samples)
    row = [c for c in range(len(cols["Data"])) if cols["Data"][c].endswith("(F_roi)ROI1.1")][0]
    e = np.abs(eul.compute().cpu().numpy()).mean(0) * 180 / np.pi
    np.testing.assert_allclose([cols["Pitch°"][row], cols["Yaw°"][row], cols["Roll°"][row]], e, rtol=1e-5)
    np.testing.assert_allclose(cols["Mean°"][row], e.mean(), rtol=1e-5)
    np.testing.assert_allclose(cols["Geodesic°"][row], geo.compute().cpu().numpy().mean() * 180 / np.pi, rtol=1e-5)
    x = xys.compute().cpu().numpy()
    np.testing.assert_allclose(cols["XY%"][row], np.sqrt((x[:, 0] ** 2 + x[:, 1] ** 2).mean()) * 100, rtol=1e-5)
    np.testing.assert_allclose(cols["S%"][row], np.sqrt((x[:, 2] ** 2).mean()) * 100, rtol=1e-5)
    bins = nme.compute()
    got = [cols[f"NME2d%_{b}"][row] for b in ("30", "60", "90", "avg")]
    want = [v * 100 for v in bins]
    np.testing.assert_allclose(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0), rtol=1e-5)  # empty yaw bins are NaN on both sides
    # text form, alignment schemes, a bare .npz as data
    S.main([ck, "--ds", os.path.join(GOLDEN, "aflw2kmini.npz"), "--roi-expansion", "1.2", "--alignment-scheme", "perspective"])
    text = capsys.readouterr().out
    assert "| Data" in text and "(stored)ROI1.2" in text and "Geodesic°" in text
    with pytest.raises(NotImplementedError):
        S.main([ck, "--vis", "rot"])
