#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE's own Python (imported from /root/reference through
oracle/tools/ref_shims.py) on torch-CPU and stores inputs' recipes + expected outputs as small .npz
fixtures under tests/golden/.

Build-container only (the reference does not exist on the GPU box).  Re-run with
    python oracle/tools/gen_golden.py
Fixtures are DATA: seeds/recipes for the inputs (oracle/synth.py regenerates them bit-for-bit from
numpy PCG64) and the reference's outputs (full for small tensors, digests for large ones).

What is pinned (SURVEY.md §8 "Caller / harness rows" table):
  model_<cfg>.npz   a2-a17,a18-a27: NetworkWithPointHead fwd (train + eval), every named per-sample
                    loss vector + weight at epochs 0/20/150 (NLL ramp), loss_sum with and without
                    dataset_weight, parameter-gradient digests, BN running statistics after 1 step.
  optim_<cfg>.npz   a28,a29: 3 steps of clip_grad_norm_(1.0) + 3-group Adam + per-epoch LambdaLR.
  schedule.npz      a28: ExponentialUpThenSteps LR-factor tables for E=200 and E=1500.
  swa.npz           a30: AveragedModel(use_buffers=True) over 3 snapshots.
  init.npz          a5: weights of a FRESH reference model under torch.manual_seed(s) - the custom conv init law
                    N(0, sqrt(2/(kh*kw*Cout))) (backbones/mobilenet_v1.py:155-158) applied in the reference's module
                    construction order (digests of every parameter and buffer, seeds 0 and 7).
"""
from __future__ import annotations

import argparse
import importlib.util
import itertools
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
import torch.nn as nn  # noqa: E402

# The training script moves loss modules with hard-coded .cuda()/.to("cuda")
# (scripts/train_poseestimator.py:204-262); on this CPU-only container these become no-ops.
nn.Module.cuda = lambda self, *a, **k: self
_orig_to = nn.Module.to
nn.Module.to = lambda self, *a, **k: self if (a and a[0] == "cuda") else _orig_to(self, *a, **k)

import trackertraincode.neuralnets.models as models  # noqa: E402
import trackertraincode.train as train  # noqa: E402
from trackertraincode.datasets.batch import Batch, Metadata  # noqa: E402
from trackertraincode.pipelines import Tag  # noqa: E402

spec = importlib.util.spec_from_file_location(
    "ref_train_script", os.path.join(ref_shims.REFERENCE_ROOT, "scripts", "train_poseestimator.py")
)
script = importlib.util.module_from_spec(spec)
spec.loader.exec_module(script)

from oracle.synth import digest, make_grads, make_inputs, make_labels, make_state  # noqa: E402

torch.set_num_threads(8)
B = 8
SPLIT = 5  # 5 x POSE_WITH_LANDMARKS + 3 x ONLY_POSE

CONFIGS = {
    # name: (model kwargs, script flags)
    "full": (
        dict(enable_point_head=True, enable_uncertainty=True, config="mobilenetv1",
             backbone_args={"use_blurpool": False}),
        dict(with_pointhead=True, with_nll_loss=True, rampup_nll_losses=True),
    ),
    "default": (  # the script's default flags: point head on, NLL off
        dict(enable_point_head=True, enable_uncertainty=False, config="mobilenetv1",
             backbone_args={"use_blurpool": False}),
        dict(with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False),
    ),
    "posonly": (
        dict(enable_point_head=False, enable_uncertainty=False, config="mobilenetv1",
             backbone_args={"use_blurpool": False}),
        dict(with_pointhead=False, with_nll_loss=False, rampup_nll_losses=False),
    ),
    "rot6d": (  # --enable-6drot: RotRepr6dWithNormalization head, Rot6dReprLoss + orthonormality constraint, NLL via as_quat()
        dict(enable_point_head=True, enable_uncertainty=True, config="mobilenetv1",
             backbone_args={"use_blurpool": False}, enable_6drot=True),
        dict(with_pointhead=True, with_nll_loss=True, rampup_nll_losses=True, enable_6drot=True),
    ),
    # --blurpool (scripts/train_poseestimator.py:294,402): BlurPool2D(3, stride 2) + stride-1 depthwise conv in the four
    # strided blocks (backbones/mobilenet_v1.py:43-55).  The two kornia helpers behind BlurPool2D (kornia is absent here) are
    # the restatements in ref_shims.py: binomial 3x3 kernel / 16, conv2d(padding 1, stride 2, groups C).
    "blurpool": (
        dict(enable_point_head=True, enable_uncertainty=False, config="mobilenetv1",
             backbone_args={"use_blurpool": True}),
        dict(with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False, with_blurpool=True),
    ),
}


def make_args(flags):
    ns = argparse.Namespace(
        backbone="mobilenetv1", batchsize=B, lr=1.0e-3, epochs=200, with_roi_train=True,
        enable_6drot=False, with_blurpool=False, swa=False,
    )
    for k, v in flags.items():
        setattr(ns, k, v)
    return ns


def build(cfgname):
    kwargs, flags = CONFIGS[cfgname]
    net = models.NetworkWithPointHead(**kwargs)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = make_state(shapes, seed=0)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return net, shapes, make_args(flags)


def make_batches(with_dataset_weight: bool):
    image, ids = make_inputs(B, seed=1234)
    lab = make_labels(B, seed=1234)
    t = lambda a: torch.from_numpy(a.copy())
    sl = [slice(0, SPLIT), slice(SPLIT, B)]
    b0 = dict(image=t(image[sl[0]]), coord_convention_id=t(ids[sl[0]]), pose=t(lab["pose"][sl[0]]),
              coord=t(lab["coord"][sl[0]]), roi=t(lab["roi"][sl[0]]), pt3d_68=t(lab["pt3d_68"][sl[0]]),
              shapeparam=t(lab["shapeparam"][sl[0]]))
    b1 = dict(image=t(image[sl[1]]), coord_convention_id=t(ids[sl[1]]), pose=t(lab["pose"][sl[1]]),
              coord=t(lab["coord"][sl[1]]), roi=t(lab["roi"][sl[1]]))
    if with_dataset_weight:
        b0["dataset_weight"] = t(lab["dataset_weight"][sl[0]])
        b1["dataset_weight"] = t(lab["dataset_weight"][sl[1]])
    return [
        Batch(Metadata(129, batchsize=SPLIT, tag=Tag.POSE_WITH_LANDMARKS), b0),
        Batch(Metadata(129, batchsize=B - SPLIT, tag=Tag.ONLY_POSE), b1),
    ]


def as_np(v):
    if hasattr(v, "value"):
        v = v.value
    return v.detach().cpu().numpy().copy()  # copy: state-dict tensors are updated in place later


def run_step(net, crit, batches, epoch):
    inputs = torch.concat([b["image"] for b in batches], dim=0)
    ids = torch.concat([b["coord_convention_id"] for b in batches], dim=0)
    preds = net(inputs, ids)
    loss_sum, all_lossvals = train.default_compute_loss(preds, batches, epoch, crit)
    by_name = train.concatenated_lossvals_by_name(itertools.chain.from_iterable(all_lossvals))
    return preds, loss_sum, by_name


def gen_model(cfgname):
    out = {}
    net, shapes, args = build(cfgname)
    out["meta"] = np.array(json.dumps({
        "config": CONFIGS[cfgname][0], "flags": CONFIGS[cfgname][1], "B": B, "split": SPLIT,
        "state_seed": 0, "input_seed": 1234, "shapes": {k: list(v) for k, v in shapes.items()},
        "bfm": "SYNTHETIC keypts/keyeigvecs (oracle/synth.py); real blob missing from reference",
        "tags": ["POSE_WITH_LANDMARKS", "ONLY_POSE"],
    }))
    train_crit, test_crit = script.setup_losses(args, net)

    # ---- eval-mode forward (adds "pose", BN uses running stats).  The synthetic running stats do
    # not match the activations (signal would die through 27 BN layers), so they are first
    # calibrated by ONE train-mode pass with momentum=1.0 (running_mean := batch mean,
    # running_var := unbiased batch var) and stored whole under calib/ (≈90 KB).
    image, ids = make_inputs(B, seed=1234)
    net.train()
    bns = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        net(torch.from_numpy(image), torch.from_numpy(ids))
    for m in bns:
        m.momentum = 0.1
    for k, v in net.state_dict().items():
        if "running_" in k:
            out[f"calib/{k}"] = as_np(v)
    net.eval()
    with torch.no_grad():
        ev = net(torch.from_numpy(image), torch.from_numpy(ids))
        ev_noid = net(torch.from_numpy(image))  # coord_convention_id=None path (models.py:340)
    for k, v in ev.items():
        out[f"eval/{k}"] = as_np(v)
    for k, v in ev_noid.items():
        out[f"eval_noid/{k}"] = as_np(v)

    # ---- backbone features + intermediates in train mode (hook)
    net.train()
    feats = {}
    h = net.convnet.register_forward_hook(lambda m, i, o: feats.update(f=o))
    batches = make_batches(False)
    for epoch in (0, 20, 150):
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_state(shapes, 0).items()})
        net.zero_grad()
        preds, loss_sum, by_name = run_step(net, train_crit, batches, epoch)
        out[f"train/e{epoch}/loss_sum"] = as_np(loss_sum)
        for name, (vals, weights) in by_name.items():
            out[f"train/e{epoch}/loss/{name}/values"] = as_np(vals)
            out[f"train/e{epoch}/loss/{name}/weights"] = as_np(weights)
        if epoch == 150:
            for k, v in preds.items():
                out[f"train/out/{k}"] = as_np(v)
            out["train/features"] = as_np(feats["f"][0])
            for i, z in enumerate(feats["f"][1]):
                out[f"train/intermediate{i}"] = digest(as_np(z))
            loss_sum.backward()
            for k, p in net.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                out[f"train/grad/{k}"] = digest(as_np(g))
            for k, v in net.state_dict().items():
                if "running_" in k or "num_batches" in k:
                    out[f"train/after/{k}"] = digest(as_np(v))
    h.remove()

    # ---- dataset_weight variant (train.py:406-411)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_state(shapes, 0).items()})
    preds, loss_sum, by_name = run_step(net, train_crit, make_batches(True), 150)
    out["train_dw/loss_sum"] = as_np(loss_sum)
    for name, (vals, weights) in by_name.items():
        out[f"train_dw/loss/{name}/weights"] = as_np(weights)

    # ---- validation criterion (scripts/train_poseestimator.py:332-338)
    sd = make_state(shapes, 0)
    sd.update({k[len("calib/"):]: v for k, v in out.items() if k.startswith("calib/")})
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    net.eval()
    with torch.no_grad():
        vb = make_batches(False)[0]
        pred = net(vb["image"])
        values = test_crit[vb.meta.tag].evaluate(pred, vb, 3)
        out["val/val_loss"] = as_np(torch.cat([(lv.val * lv.weight) for lv in values]).sum())
        out["val/names"] = np.array(json.dumps([lv.name for lv in values]))
    np.savez_compressed(os.path.join(GOLD, f"model_{cfgname}.npz"), **out)
    print(cfgname, "model:", len(out), "entries")


def gen_optim(cfgname):
    out = {}
    net, shapes, args = build(cfgname)
    args.epochs = 20  # n_up = 2, step at epoch 10: all three LR regimes inside few "epochs"
    train_crit, _ = script.setup_losses(args, net)
    optimizer, scheduler = script.create_optimizer(net, args)
    out["meta"] = np.array(json.dumps({
        "epochs": args.epochs, "lr": args.lr, "steps": 3, "clip": 1.0,
        "group_sizes": [len(g["params"]) for g in optimizer.param_groups],
        "note": "one optimiser step per 'epoch': scheduler.step() after every step",
    }))
    net.train()
    batches = make_batches(False)
    for step in range(3):
        optimizer.zero_grad()
        _, loss_sum, _ = run_step(net, train_crit, batches, step)
        loss_sum.backward()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        out[f"step{step}/loss_sum"] = as_np(loss_sum)
        out[f"step{step}/grad_norm"] = as_np(gn)
        out[f"step{step}/lrs"] = np.array([g["lr"] for g in optimizer.param_groups])
        optimizer.step()
        scheduler.step()
    for k, v in net.state_dict().items():
        out[f"final/{k}"] = digest(as_np(v))

    # ---- the same optimiser/schedule driven by FIXED synthetic gradients (no fwd/bwd in the loop,
    # so the result is reproducible to rounding): pins clip + Adam + LR schedule arithmetic sharply.
    net2, shapes, args2 = build(cfgname)
    args2.epochs = 20
    optimizer, scheduler = script.create_optimizer(net2, args2)
    pshapes = {k: tuple(p.shape) for k, p in net2.named_parameters()}
    for step, gscale in enumerate((1.0e-3, 1.0e-4, 1.0e-2)):  # norms ~1.8 (clipped), ~0.18 (not), ~18
        g = make_grads(pshapes, seed=200 + step, scale=gscale)
        for k, p in net2.named_parameters():
            p.grad = torch.from_numpy(g[k].copy())
        gn = torch.nn.utils.clip_grad_norm_(net2.parameters(), 1.0)
        out[f"fixed/step{step}/grad_norm"] = as_np(gn)
        optimizer.step()
        scheduler.step()
    for k, v in net2.state_dict().items():
        out[f"fixed/final/{k}"] = digest(as_np(v))
    np.savez_compressed(os.path.join(GOLD, f"optim_{cfgname}.npz"), **out)
    print(cfgname, "optim:", len(out), "entries")


def gen_schedule():
    out = {}
    for E in (200, 1500):
        lin = nn.Linear(1, 1)
        opt = torch.optim.Adam(lin.parameters(), lr=1.0)
        sch = train.ExponentialUpThenSteps(opt, max(1, E // 10), 0.1, [E // 2])
        f = []
        for _ in range(E):
            f.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[f"E{E}"] = np.array(f)
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **out)


def gen_swa():
    from torch.optim.swa_utils import AveragedModel

    out = {}
    m = nn.Sequential(nn.Conv2d(1, 4, 3, bias=False), nn.BatchNorm2d(4))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    swa = AveragedModel(m, device="cpu", use_buffers=True)
    for s in range(3):
        sd = make_state({("bn." + k if k.startswith("1.") else k): v for k, v in shapes.items()}, seed=100 + s)
        sd = {k[3:] if k.startswith("bn.") else k: v for k, v in sd.items()}
        sd["1.num_batches_tracked"] = np.array(s + 1, dtype=np.int64)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd.items()})
        swa.update_parameters(m)
    for k, v in swa.module.state_dict().items():
        out[k] = as_np(v)
    out["n_averaged"] = as_np(swa.n_averaged)
    np.savez_compressed(os.path.join(GOLD, "swa.npz"), **out)


def gen_init():
    """Seed parity of fresh models: torch's CPU generator + the reference's construction order decide every value."""
    out = {}
    for seed in (0, 7):
        torch.manual_seed(seed)
        net = models.NetworkWithPointHead(enable_point_head=True, enable_uncertainty=True, config="mobilenetv1",
                                          backbone_args={"use_blurpool": False})
        for k, v in net.state_dict().items():
            if k.endswith("keypts") or k.endswith("keyeigvecs"):
                continue  # the 3DMM blob is absent from the reference checkout: synthetic, not part of the init law
            out[f"seed{seed}/{k}"] = digest(as_np(v).astype(np.float64))
    out["meta"] = json.dumps({"seeds": [0, 7], "config": dict(enable_point_head=True, enable_uncertainty=True, config="mobilenetv1",
                                                               backbone_args={"use_blurpool": False})})
    np.savez_compressed(os.path.join(GOLD, "init.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["model", "optim", "schedule", "swa", "init"]
    if "init" in which:
        gen_init()
    if "schedule" in which:
        gen_schedule()
    if "swa" in which:
        gen_swa()
    only = [w[4:] for w in which if w.startswith("cfg=")]
    for cfg in CONFIGS:
        if only and cfg not in only:
            continue
        if "model" in which:
            gen_model(cfg)
        if "optim" in which and cfg in ("full", "default"):
            gen_optim(cfg)
    print("done")
