#!/usr/bin/env python3
"""Golden vectors of the export surface: runs the REFERENCE's own scripts/export_model.py wrappers - imported from
/root/reference through oracle/tools/ref_shims.py plus name-only stand-ins for the absent onnx / onnxsim / onnxruntime /
onnxconverter_common packages (none takes part in the wrappers' arithmetic) - and stores in tests/golden/export_contract.npz:

  * ModelForOpenTrack / ExportModel: output names, order and values of the eval-mode network on seeded inputs and weights
    (configs "default" and "full" = with uncertainty heads), reference scripts/export_model.py:116-169;
  * clear_denormals on a state dict with magnitudes 1e-30 .. 1e-10 around the 1e-20 threshold, :36-50.

Build-container only.  Re-run with  python oracle/tools/gen_golden_export.py
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import ref_shims  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
torch = ref_shims.install(gmm_npz=os.path.join(GOLD, "shapeparams_gmm.npz"))
for name in ("onnx", "onnx.shape_inference", "onnxsim", "onnxruntime", "onnxconverter_common", "onnxconverter_common.float16"):
    ref_shims._stub(name)
import trackertraincode.neuralnets.models as models  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_export_script", os.path.join(ref_shims.REFERENCE_ROOT, "scripts", "export_model.py"))
script = importlib.util.module_from_spec(spec)
spec.loader.exec_module(script)

from oracle.synth import make_inputs, make_state  # noqa: E402

out = {}
meta = {"configs": {}, "input_seed": 4321, "state_seed": 0}
for cfg, unc in (("default", False), ("full", True)):
    net = models.NetworkWithPointHead(enable_point_head=True, enable_uncertainty=unc, config="mobilenetv1", backbone_args={"use_blurpool": False})
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = make_state(shapes, seed=meta["state_seed"])
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    net.eval()
    info = {}
    for kind, B in (("opentrack", 1), ("complete", 3)):
        wrapper = script.ModelForOpenTrack(net) if kind == "opentrack" else script.ExportModel(net)
        wrapper.eval()
        image, _ = make_inputs(B, seed=meta["input_seed"])
        with torch.no_grad():
            ys = wrapper(torch.from_numpy(image))
        names = list(wrapper.output_names)
        vals = [getattr(y, "value", y) for y in ys]
        info[kind] = {"input_names": list(wrapper.input_names), "output_names": names, "B": B, "input_resolution": int(wrapper.input_resolution)}
        for n, v in zip(names, vals):
            out[f"{cfg}/{kind}/{n}"] = v.detach().numpy()
    meta["configs"][cfg] = info
# clear_denormals
rng = np.random.default_rng(11)
mags = 10.0 ** rng.uniform(-30, -10, size=(64, 7))
probe = {"a.weight": (mags * rng.choice([-1.0, 1.0], size=mags.shape)).astype(np.float32), "b.bias": np.array([0.0, 1e-20, -1e-20, 1.0000001e-20, 3e-39, 1.0], np.float32),
         "c.num_batches_tracked": np.array(7, np.int64)}
cleared = script.clear_denormals({k: torch.from_numpy(v.copy()) for k, v in probe.items()})
for k, v in probe.items():
    out[f"denormals/in/{k}"] = v
    out[f"denormals/out/{k}"] = cleared[k].numpy()
out["meta"] = np.array(json.dumps(meta))
np.savez_compressed(os.path.join(GOLD, "export_contract.npz"), **out)
print(json.dumps(meta, indent=1))
print({k: v.shape for k, v in out.items() if k != "meta"})
