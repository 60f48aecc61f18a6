"""CPU tests of the host-side surface: C-ABI exports, state-dict inventory, checkpoint io, Batch
collation, CPU eval/export path against the reference's golden outputs, schedules."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.synth import make_inputs, make_state
from util import GOLDEN, PKG, REPO, build_net, load_golden


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    lib_path = os.path.join(PKG, "libttk_hip.so")
    assert os.path.exists(lib_path), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(REPO, "include", "ttk.h")).read()
    declared = sorted(set(re.findall(r"\b(ttk_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) > 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ttk.h but not exported"
    import trackertraincode._hip as H

    assert set(H.exported_symbols()) <= set(declared)
    lib.ttk_abi_version.restype = ctypes.c_int
    assert lib.ttk_abi_version() == H.ABI_VERSION


@pytest.mark.parametrize("unc,pt", [(True, True), (False, True), (False, False)])
def test_state_dict_inventory(unc, pt):
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    net = NetworkWithPointHead(enable_point_head=pt, enable_uncertainty=unc)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = R.state_shapes(pt, unc)  # pinned to the reference by test_oracle_golden
    assert list(mine) == list(ref) and mine == ref
    assert net.get_config() == {"enable_point_head": pt, "enable_face_detector": False, "config": "mobilenetv1",
                                "enable_uncertainty": unc, "use_local_pose_offset": True, "backbone_args": {}, "enable_6drot": False}
    assert net.name == "NetworkWithPointHead_mobilenetv1" and net.input_resolution == 129 and net.input_resolutions == (129,)


def test_resnet18_blurpool_state_dict_inventory():
    """resnet18(use_blurpool=True): the reference's CustomBlock / BlurPool2D key names (resnet.py:31-49,63-66) = the oracle's inventory."""
    from trackertraincode.backbones.resnet import resnet18

    net = resnet18(use_blurpool=True)
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = R.resnet18_state_shapes(use_blurpool=True)
    assert list(mine) == list(ref) and mine == ref
    assert "layers.3.kernel" in mine and "layers.4.0.conv1.0.kernel" in mine and "layers.7.1.conv1.1.weight" in mine
    assert torch.equal(net.layers[3].kernel, torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0)


def test_state_dict_inventory_blurpool():
    """--blurpool: conv_dw becomes Sequential(BlurPool2D, Conv2d) in the four strided blocks (reference mobilenet_v1.py:43-55)."""
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    _, meta = load_golden("model_blurpool.npz")
    net = NetworkWithPointHead(**meta["config"])
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = {k: tuple(v) for k, v in meta["shapes"].items()}
    assert list(mine) == list(ref) and mine == ref
    assert net.get_config()["backbone_args"] == {"use_blurpool": True}
    k = net.convnet.dw2_2.conv_dw[0].kernel
    assert torch.equal(k, torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0)


@pytest.mark.parametrize("cfg", ["full", "posonly", "blurpool"])
def test_cpu_eval_export_path_matches_reference_golden(cfg):
    d, meta = load_golden(f"model_{cfg}.npz")
    cal = {k[len("calib/"):]: d[k] for k in d.files if k.startswith("calib/")}
    net = build_net(meta, "cpu", cal).eval()
    image, ids = make_inputs(meta["B"], seed=meta["input_seed"])
    with torch.no_grad():
        out = net(torch.from_numpy(image), torch.from_numpy(ids))
    for k in [k[len("eval/"):] for k in d.files if k.startswith("eval/")]:
        v = out[k].value if hasattr(out[k], "value") else out[k]
        np.testing.assert_allclose(v.numpy(), d["eval/" + k], rtol=2e-4, atol=2e-5, err_msg=k)
    with pytest.raises(RuntimeError, match="CUDA"):
        net.train()(torch.from_numpy(image))


def test_checkpoint_roundtrip(tmp_path):
    from trackertraincode.neuralnets import models

    net = models.NetworkWithPointHead(enable_point_head=False, enable_uncertainty=True)
    f = str(tmp_path / "m.ckpt")
    models.save_model(net, f)
    raw = torch.load(f, weights_only=True)
    assert set(raw) == {"state_dict", "class_name", "config"} and raw["class_name"] == "NetworkWithPointHead"
    net2 = models.load_model(f)
    for (k1, v1), (k2, v2) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_batch_collation():
    from trackertraincode.datasets.batch import Batch, Metadata

    frames = [Batch(Metadata(129, 0, tag="a"), x=torch.full((2,), float(i))) for i in range(3)]
    frames += [Batch(Metadata(129, 2, tag="b"), x=torch.zeros(2, 2))]
    out = Batch.Collation(lambda b: b.meta.tag)(frames)
    assert [b.meta.tag for b in out] == ["a", "b"] and out[0]["x"].shape == (3, 2) and out[0].meta.batchsize == 3
    assert out[1].meta.batchsize == 2 and out[0].meta.prefixshape == (3,)
    vids = [Batch(Metadata(129, 0, tag="v", seq=[0, 2]), x=torch.zeros(2, 1)), Batch(Metadata(129, 0, tag="v", seq=[0, 3]), x=torch.ones(3, 1))]
    v = Batch.collate(vids)
    assert v.meta.seq == [0, 2, 5] and v.meta.batchsize == 2 and v["x"].shape == (5, 1) and v.meta.prefixshape == (5,)
    assert [f["x"].item() for f in v.iter_frames()] == [0, 0, 1, 1, 1]
    assert [s["x"].shape[0] for s in v.iter_sequences()] == [2, 3]


def test_lr_schedule_matches_reference_tables():
    import trackertraincode.train as train

    s = np.load(os.path.join(GOLDEN, "schedule.npz"))
    for E in (200, 1500):
        lin = torch.nn.Linear(1, 1)
        opt = torch.optim.SGD(lin.parameters(), lr=1.0)
        sch = train.ExponentialUpThenSteps(opt, max(1, E // 10), 0.1, [E // 2])
        f = []
        for _ in range(E):
            f.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        np.testing.assert_allclose(np.array(f), s[f"E{E}"], rtol=1e-12)


def test_swa_callback_matches_reference():
    import trackertraincode.train as train

    s = np.load(os.path.join(GOLDEN, "swa.npz"))
    m = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, bias=False), torch.nn.BatchNorm2d(4))
    cb = train.SwaCallback(start_epoch=-1)
    cb.on_train_start(m)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    for i in range(3):
        sd = make_state({("bn." + k if k.startswith("1.") else k): v for k, v in shapes.items()}, seed=100 + i)
        sd = {(k[3:] if k.startswith("bn.") else k): torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        sd["1.num_batches_tracked"] = torch.tensor(i + 1)
        m.load_state_dict(sd)
        cb.on_train_epoch_end(i, m)
    for k, v in cb.swa_model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), s[k], rtol=1e-6, atol=1e-7, err_msg=k)


def test_resnet18_module_surface():
    """State-dict inventory (torchvision naming, reference resnet.py:68-73), init laws and the CPU eval path."""
    import torch
    from oracle import refmodel as R
    from trackertraincode.backbones.resnet import BasicBlock, resnet18

    net = resnet18()
    assert net.num_features == 512
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == R.resnet18_state_shapes()
    assert list(net.state_dict().keys()) == list(R.resnet18_state_shapes().keys())
    assert all(float(m.bn2.weight.abs().max()) == 0.0 for m in net.modules() if isinstance(m, BasicBlock))  # zero_init_residual
    net.eval()
    x = torch.rand(2, 1, 129, 129) - 0.5
    with torch.no_grad():
        feat, inter = net(x)
        st = {k: v.clone() for k, v in net.state_dict().items()}
        ref, _ = R.resnet18_forward(st, x, False)
    assert inter is None and feat.shape == (2, 512)
    torch.testing.assert_close(feat, ref, rtol=1e-5, atol=1e-6)
    blur = resnet18(use_blurpool=True).eval()  # the CPU module path of the blur variant against the oracle's restatement
    with torch.no_grad():
        feat_b, _ = blur(x)
        ref_b, _ = R.resnet18_forward({k: v.clone() for k, v in blur.state_dict().items()}, x, False)
    torch.testing.assert_close(feat_b, ref_b, rtol=1e-5, atol=1e-6)


def test_eval_metrics_and_rotation_conventions():
    """trackertraincode.utils / eval metrics against their definitions (reference utils.py:41-64, eval.py:337-440)."""
    import numpy as np
    import torch
    from scipy.spatial.transform import Rotation

    from trackertraincode import eval as E
    from trackertraincode import utils

    rng = np.random.default_rng(0)
    pyr = rng.uniform(-1.2, 1.2, (32, 3))
    rot = utils.aflw_rotation_conversion(pyr[:, 0], pyr[:, 1], pyr[:, 2])
    back = np.array([utils.inv_aflw_rotation_conversion(r) for r in rot])
    np.testing.assert_allclose(back, pyr, atol=1e-9)  # the two conversions invert each other
    q = torch.from_numpy(rot.as_quat().astype(np.float32))
    d = rng.uniform(-0.05, 0.05, (32, 3))
    q2 = torch.from_numpy(utils.aflw_rotation_conversion(*(pyr + d).T).as_quat().astype(np.float32))
    m = E.EulerAngleErrors()
    m.update({"pose": q2}, {"pose": q})
    np.testing.assert_allclose(m.compute().numpy(), np.abs(d), atol=2e-5)
    geo = E.GeodesicError()
    geo.update({"pose": q2}, {"pose": q})
    ref = (Rotation.from_quat(q.numpy()).inv() * Rotation.from_quat(q2.numpy())).magnitude()
    np.testing.assert_allclose(geo.compute().numpy(), ref, atol=2e-5)
    tab = E.pose_error_table(m.compute(), geo.compute())
    assert abs(tab["mae"] - np.abs(d).mean() * 180 / np.pi) < 1e-3
    # landmark NME: a pure in-plane shift of every point by 1 % of the box size
    gt = torch.from_numpy(rng.uniform(0, 100, (4, 68, 3)).astype(np.float32))
    size = torch.sqrt((gt[:, :, 0].amax(1) - gt[:, :, 0].amin(1)) * (gt[:, :, 1].amax(1) - gt[:, :, 1].amin(1)))
    pred = gt.clone()
    pred[:, :, 0] += 0.01 * size[:, None]
    nme = E.UnweightedKptNME()
    nme.update({"pt3d_68": pred}, {"pt3d_68": gt})
    np.testing.assert_allclose(nme.compute().numpy(), 0.01, rtol=1e-4)
    k = E.KptNME()
    yaw = np.array([10.0, 40.0, 70.0, 20.0]) * np.pi / 180
    qq = torch.from_numpy(utils.aflw_rotation_conversion(np.zeros(4), yaw, np.zeros(4)).as_quat().astype(np.float32))
    k.update({"pt3d_68": pred}, {"pt3d_68": gt, "pose": qq})
    res = k.compute()
    assert abs(res.bin_30_nme - 0.01) < 1e-5 and abs(res.bin_60_nme - 0.01) < 1e-5 and abs(res.avg_nme - 0.01) < 1e-5


def test_conv_bn_fusion_keeps_the_eval_output():
    """neuralnets/bnfusion.fuse_convbn on the backbone's plain-torch eval path (reference bnfusion.py:24-63)."""
    import torch.fx as fx
    import torch.nn as nn
    from trackertraincode.backbones.mobilenet_v1 import MobileNet
    from trackertraincode.neuralnets.bnfusion import fuse_convbn, torch_eval_module

    torch.manual_seed(0)
    net = MobileNet(num_classes=None) if "num_classes" in MobileNet.__init__.__code__.co_varnames else MobileNet()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    net.eval()
    with pytest.raises(RuntimeError):
        torch_eval_module(MobileNet().train())
    gm = fx.symbolic_trace(torch_eval_module(net))
    fused = fuse_convbn(gm)
    assert sum(isinstance(m, nn.BatchNorm2d) for m in fused.modules()) == 0
    assert sum(isinstance(m, nn.BatchNorm2d) for m in gm.modules()) == 27  # the input graph is left alone
    x = torch.rand(2, 1, 129, 129) - 0.5
    with torch.no_grad():
        a, b = gm(x), fused(x)
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_prepare_finetune_freezes_norm_layers_like_the_reference():
    """models.py:378-394: prepare_finetune() returns one parameter group per backbone sub-module (+ the rest), and train() then
    keeps every normalisation layer of the backbone in eval mode with frozen affine parameters (modelcomponents.py:208-215)."""
    import torch
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    net = NetworkWithPointHead(enable_point_head=False, enable_uncertainty=False)
    groups = net.prepare_finetune()
    flat = [p for g in groups for p in g]
    assert len(flat) == len(set(map(id, flat))) == len(list(net.parameters()))
    assert len(groups) == 13 * 5 + 1  # conv_dw, bn_dw, conv_sep, bn_sep, relu of the 13 blocks; everything else
    assert net.train() is net and net.training
    bns = [m for m in net.convnet.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(bns) == 27 and not any(m.training for m in bns)
    assert not any(p.requires_grad for m in bns for p in m.parameters())
    assert all(p.requires_grad for n, p in net.named_parameters() if ".bn" not in n)
    net.eval()
    assert not net.training
