#!/usr/bin/env python3
"""Golden vectors of the evaluation TABLE (the part around the metrics), produced by the reference's own code imported
through ref_shims (build container only):

  trackertraincode/eval.py:443-483   compute_mean_rotation, compute_opal_paper_alignment (per-individual alignment of the OPAL paper)
  trackertraincode/eval.py:485-544   PerspectiveCorrector.corrected_rotation
  trackertraincode/eval.py:547-600   AlignedRotationErrorMetric.compute - the composition of the above with _aflw3d_euler_errors /
                                     geodesicdistance (torchmetrics is absent: the generator calls the same functions in compute()'s order)
  scripts/evaluate_pose_network.py:45-66,107-193   RoiConfig.__str__, comprehensive_roi_configs, TableBuilder (github table and JSON)
  scripts/evaluate_pose_network.py:205-291          report(): the row arithmetic (mean |euler| in degrees, geodesic, RMSE of position and size
                                     in percent, NME columns), run with the data / network / predictor replaced by fixed metric outputs

-> tests/golden/eval_table.npz
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shims  # noqa: E402

torch = ref_shims.install(synthetic_bfm=True)
import trackertraincode.eval as RE  # noqa: E402
from trackertraincode.neuralnets import torchquaternion  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_eval_script", os.path.join(ref_shims.REFERENCE_ROOT, "scripts", "evaluate_pose_network.py"))
S = importlib.util.module_from_spec(spec)
spec.loader.exec_module(S)

rng = np.random.default_rng(2718)
out = {}


def unit(q):
    return (q / np.linalg.norm(q, axis=-1, keepdims=True)).astype(np.float32)


# ---------------------------------------------------------------- alignment schemes
B = 40
q_t = unit(np.concatenate([rng.standard_normal((B, 3)) * 0.35, np.ones((B, 1))], -1))
ids = np.repeat(np.arange(4), B // 4).astype(np.int32)
bias = unit(np.concatenate([rng.standard_normal((4, 3)) * 0.06, np.ones((4, 1))], -1))  # a systematic offset per individual
noise = unit(np.concatenate([rng.standard_normal((B, 3)) * 0.02, np.ones((B, 1))], -1))
q_p = torchquaternion.mult(torchquaternion.mult(torch.from_numpy(q_t), torch.from_numpy(bias[ids])), torch.from_numpy(noise)).numpy().astype(np.float32)
coord_p = np.stack([rng.uniform(50, 590, B), rng.uniform(40, 440, B), rng.uniform(40, 90, B)], -1).astype(np.float32)
sizes_hw = np.tile(np.array([[480, 640]], dtype=np.int64), (B, 1))
out.update(al_pose_target=q_t, al_pose_pred=q_p, al_individual=ids, al_coord_pred=coord_p, al_image_hw=sizes_hw, al_fov=np.float64(S.BIWI_HORIZONTAL_FOV))
from scipy.spatial.transform import Rotation  # noqa: E402

out["mean_rotation_quat"] = RE.compute_mean_rotation(Rotation.from_quat(q_t).inv() * Rotation.from_quat(q_p)).as_quat()
aligned = RE.compute_opal_paper_alignment(torch.from_numpy(q_p), torch.from_numpy(q_t), ids)
out["opal_aligned"] = aligned.numpy()
image_wh = torch.flip(torch.from_numpy(sizes_hw), dims=(-1,))  # compute(): "Format to WH"
persp = RE.PerspectiveCorrector(S.BIWI_HORIZONTAL_FOV).corrected_rotation(image_wh, torch.from_numpy(coord_p), torch.from_numpy(q_p))
out["perspective_corrected"] = persp.numpy()
for mode, q in (("opal23", aligned), ("perspective", persp)):
    out[f"aligned_euler_{mode}"] = RE._aflw3d_euler_errors(q, torch.from_numpy(q_t)).numpy()
    out[f"aligned_geo_{mode}"] = torchquaternion.geodesicdistance(q, torch.from_numpy(q_t)).numpy()

# ---------------------------------------------------------------- RoiConfig / TableBuilder / report()
out["roi_config_names"] = np.array(json.dumps([str(c) for c in S.comprehensive_roi_configs] + [str(S.RoiConfig()), str(S.RoiConfig(1.3, True))]))
out["roi_config_fields"] = np.array(json.dumps([list(c) for c in S.comprehensive_roi_configs]))

n = 30
res = {"pose_errs": (rng.standard_normal((n, 3)) * 0.03).astype(np.float32), "geodesic_errs": rng.uniform(0.01, 0.2, n).astype(np.float32),
       "euler_errs": (rng.standard_normal((n, 3)) * 0.08).astype(np.float32), "uw_nme_3d": rng.uniform(0.02, 0.08, n).astype(np.float32)}
nme_2d = RE.KptNmeResults(0.031, 0.044, 0.062, float(np.average([0.031, 0.044, 0.062])))
for k, v in res.items():
    out["res_" + k] = v
out["res_nme_2d"] = np.array(list(nme_2d))


class _FakePredictor:
    def __init__(self, net, expansion):
        pass

    calls = 0

    def evaluate(self, metrics, loader):
        f = np.float32(1.0 + 0.125 * _FakePredictor.calls)  # a different row for every (model, data, roi config): results * (1 + call / 8)
        _FakePredictor.calls += 1
        r = {k: torch.from_numpy(v * f) for k, v in res.items() if k in metrics.names or k == "pose_errs"}
        if "nme_2d" in metrics.names:
            r["nme_2d"] = RE.KptNmeResults(*[float(x * f) for x in nme_2d])
        return r


class _FakeCollection:
    def __init__(self, d):
        self.names = set(d)

    def add_metrics(self, d):
        self.names |= set(d)


# torchmetrics is absent (its Metric base is a name-only stand-in): report() only constructs the metric objects and hands them to the
# collection, so the constructors' add_state() calls are given a no-op
RE._SimpleConcatenatingErrorMetric.add_state = lambda self, *a, **k: None
RE.KptNME.add_state = lambda self, *a, **k: None
S.eval.Predictor = _FakePredictor
S.torchmetrics.MetricCollection = _FakeCollection
S.trackertraincode.pipelines.make_validation_loader = lambda *a, **k: None
S.load_pose_network = lambda fn, dev: None
tables = {}
for with_points in (True, False):
    S.compute_pred_keys = lambda loader, net, _p=with_points: ["coord", "pose", "roi"] + (["pt3d_68"] if _p else [])
    tb = S.TableBuilder()
    _FakePredictor.calls = 0
    args = types.SimpleNamespace(alignment_scheme="none", device="cpu", vis="none")
    for fn in ("/models/run1/best.ckpt", "/models/run2/best.ckpt"):
        for ds in ("aflw2k3d", "biwi"):
            for cfg in (S.RoiConfig(), S.RoiConfig(1.2, False, False)):
                S.report(fn, ds, cfg, args, tb)
    tables[f"table_points{int(with_points)}"] = tb.build()
    try:
        tables[f"json_points{int(with_points)}"] = tb.build_json()
    except TypeError as e:  # with landmark columns the reference hands numpy float32 scalars to json.dumps: "--json" raises there
        tables[f"json_points{int(with_points)}"] = "TypeError: " + str(e)
for k, v in tables.items():
    out[k] = np.array(v)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "eval_table.npz"), **out)
print(tables["table_points1"])
print("eval_table.npz:", len(out), "entries")
