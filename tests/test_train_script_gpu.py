"""scripts/train_poseestimator.py run as a program (one process, the MI355X) over its flags: the paths the other tests enter through
functions are entered through `main()` here - argument parsing, loaders, network / loss / optimiser construction, fit with validation,
SWA, checkpoints.  Epochs are cut to six steps."""
import os
import shutil
import subprocess
import sys

import pytest
import torch

from util import GOLDEN

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WRAP = r"""
import sys, os, runpy
sys.argv = [sys.argv[1]] + sys.argv[2:]
import trackertraincode.pipelines as P
_orig = P.make_pose_estimation_loaders
def short(*a, **k):
    if k.get("datasets") != "synthetic":  # the bundled 16-frame file: frames 8.. train, 0..7 validate
        P._POSE_SHARDS[P.Id.AFLW2k3d] = ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, 1000.0, (8, None))
        P._TEST_SHARD = ("aflw2k", P.Tag.POSE_WITH_LANDMARKS, (0, 8))
        k["steps_per_epoch"] = 6
    tr, te, n = _orig(*a, **k)
    if hasattr(tr, "_steps"):
        tr._steps = 6
    return tr, te, n
P.make_pose_estimation_loaders = short
runpy.run_path(sys.argv[0], run_name="__main__")
"""
CASES = {
    "defaults": (["--ds", "synthetic", "--batchsize", "32", "--epochs", "2"], "NetworkWithPointHead_mobilenetv1"),
    "nll_6drot_swa": (["--ds", "synthetic", "--batchsize", "32", "--epochs", "3", "--with-nll-loss", "--rampup-nll-losses", "--enable-6drot", "--with-swa"],
                      "NetworkWithPointHead_mobilenetv1"),
    "resnet18_posonly": (["--ds", "synthetic", "--batchsize", "16", "--epochs", "1", "--backbone", "resnet18", "--no-pointhead", "--no-imgaug"],
                         "NetworkWithPointHead_resnet18"),
    "resnet18_blurpool": (["--ds", "synthetic", "--batchsize", "16", "--epochs", "1", "--backbone", "resnet18", "--blurpool"], "NetworkWithPointHead_resnet18"),
    "blurpool_graph": (["--ds", "synthetic", "--batchsize", "32", "--epochs", "2", "--blurpool", "--graph-steps"],
                       "NetworkWithPointHead_mobilenetv1"),
    "graph_nll_bf16_compute": (["--ds", "synthetic", "--batchsize", "32", "--epochs", "2", "--graph-steps", "--with-nll-loss", "--rampup-nll-losses", "--precision", "bf16-compute"],
                               "NetworkWithPointHead_mobilenetv1"),
    "shards_landmark_roi": (["--ds", "aflw2k:500", "--batchsize", "8", "--epochs", "2", "--roi-override", "landmarks", "--ds-weighting", "--raug", "20"],
                            "NetworkWithPointHead_mobilenetv1"),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_script_main(case, tmp_path):
    from trackertraincode.neuralnets.models import load_model

    flags, name = CASES[case]
    script = os.path.join(REPO, "neuralnet-tracker-traincode_amd", "scripts", "train_poseestimator.py")
    wrap = tmp_path / "wrap.py"
    wrap.write_text(WRAP)
    data = tmp_path / "data"
    data.mkdir()
    shutil.copy(os.path.join(GOLDEN, "aflw2kmini.npz"), data / "aflw2k.npz")
    env = dict(os.environ, DATADIR=str(data), PYTHONPATH=os.path.join(REPO, "neuralnet-tracker-traincode_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(wrap), script, *flags, "--outdir", str(tmp_path / "out")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    files = sorted(os.listdir(tmp_path / "out" / name))
    assert "last.ckpt" in files and "best.ckpt" in files, files
    if "--with-swa" in flags:
        assert any("swa" in f for f in files), files
    net = load_model(str(tmp_path / "out" / name / "last.ckpt"))  # the plain {state_dict, class_name, config} format
    assert all(torch.isfinite(v).all() for v in net.state_dict().values() if v.is_floating_point())
    cfg = net.get_config()
    assert cfg["config"] == ("resnet18" if "resnet18" in flags else "mobilenetv1")
    assert cfg["enable_point_head"] == ("--no-pointhead" not in flags) and cfg["enable_uncertainty"] == ("--with-nll-loss" in flags)
    assert cfg["enable_6drot"] == ("--enable-6drot" in flags) and cfg["backbone_args"] == {"use_blurpool": "--blurpool" in flags}
