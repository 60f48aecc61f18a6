"""Pins the CPU oracle (oracle/refmodel.py) against the golden vectors produced by importing the
reference itself (oracle/tools/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.synth import digest_close, make_grads, make_inputs, make_labels, make_state

torch.set_num_threads(min(8, os.cpu_count() or 1))
CFGS = ["full", "default", "posonly", "rot6d", "blurpool"]


def _load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    meta = json.loads(str(d["meta"]))
    return d, meta


def _batches(meta, with_dw=False):
    B, split = meta["B"], meta["split"]
    lab = make_labels(B, seed=meta["input_seed"])
    t = lambda a: torch.from_numpy(a.copy())
    b0 = {k: t(lab[k][:split]) for k in ("pose", "coord", "roi", "pt3d_68", "shapeparam")}
    b1 = {k: t(lab[k][split:]) for k in ("pose", "coord", "roi")}
    if with_dw:
        b0["dataset_weight"], b1["dataset_weight"] = t(lab["dataset_weight"][:split]), t(lab["dataset_weight"][split:])
    b0.update(tag="POSE_WITH_LANDMARKS", n=split)
    b1.update(tag="ONLY_POSE", n=B - split)
    return [b0, b1]


def _criterions(meta, golden_dir):
    fl = meta["flags"]
    gmm = R.ShapeGmm(os.path.join(golden_dir, "shapeparams_gmm.npz"))
    return R.setup_losses(with_pointhead=fl["with_pointhead"], with_nll_loss=fl["with_nll_loss"],
                          rampup_nll_losses=fl["rampup_nll_losses"], epochs=200, gmm=gmm,
                          enable_6drot=fl.get("enable_6drot", False))


@pytest.mark.parametrize("cfg", CFGS)
def test_state_inventory_matches_reference(cfg, golden_dir):
    d, meta = _load(golden_dir, f"model_{cfg}.npz")
    mine = R.state_shapes(meta["config"]["enable_point_head"], meta["config"]["enable_uncertainty"],
                          enable_6drot=meta["config"].get("enable_6drot", False),
                          use_blurpool=meta["config"]["backbone_args"]["use_blurpool"])
    ref = {k: tuple(v) for k, v in meta["shapes"].items()}
    assert list(mine.keys()) == list(ref.keys())  # same names, same order (checkpoint surface)
    assert mine == ref


@pytest.mark.parametrize("cfg", CFGS)
def test_eval_forward(cfg, golden_dir):
    d, meta = _load(golden_dir, f"model_{cfg}.npz")
    sd = make_state({k: tuple(v) for k, v in meta["shapes"].items()}, meta["state_seed"])
    sd.update({k[len("calib/"):]: d[k] for k in d.files if k.startswith("calib/")})
    st = R.state_from_numpy(sd, requires_grad=False)
    image, ids = make_inputs(meta["B"], seed=meta["input_seed"])
    with torch.no_grad():
        out, _ = R.network_forward(st, torch.from_numpy(image), torch.from_numpy(ids), meta["config"], False)
        out_noid, _ = R.network_forward(st, torch.from_numpy(image), None, meta["config"], False)
    for prefix, o in (("eval/", out), ("eval_noid/", out_noid)):
        keys = [k[len(prefix):] for k in d.files if k.startswith(prefix)]
        assert set(keys) == set(o.keys())
        for k in keys:
            np.testing.assert_allclose(o[k].numpy(), d[prefix + k], rtol=2e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("cfg", CFGS)
def test_train_step_losses_and_grads(cfg, golden_dir):
    d, meta = _load(golden_dir, f"model_{cfg}.npz")
    shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
    crit, _ = _criterions(meta, golden_dir)
    image, ids = make_inputs(meta["B"], seed=meta["input_seed"])
    for epoch in (0, 20, 150):
        st = R.state_from_numpy(make_state(shapes, meta["state_seed"]))
        out, feat = R.network_forward(st, torch.from_numpy(image), torch.from_numpy(ids), meta["config"], True)
        loss_sum, by_name = R.compute_loss(out, _batches(meta), epoch, crit)
        names = [k.split("/")[3] for k in d.files if k.startswith(f"train/e{epoch}/loss/") and k.endswith("/values")]
        assert list(by_name.keys()) == names  # same loss names in the same order
        for n in names:
            np.testing.assert_allclose(by_name[n][0].detach().numpy(), d[f"train/e{epoch}/loss/{n}/values"], rtol=3e-4, atol=3e-5, err_msg=n)
            np.testing.assert_allclose(by_name[n][1].detach().numpy(), d[f"train/e{epoch}/loss/{n}/weights"], rtol=1e-6, atol=0, err_msg=n)
        np.testing.assert_allclose(loss_sum.item(), d[f"train/e{epoch}/loss_sum"], rtol=1e-4)
    # epoch 150: outputs, features, grads, BN buffers
    for k in [k for k in d.files if k.startswith("train/out/")]:
        np.testing.assert_allclose(out[k[len("train/out/"):]].detach().numpy(), d[k], rtol=3e-4, atol=3e-5, err_msg=k)
    np.testing.assert_allclose(feat.detach().numpy(), d["train/features"], rtol=2e-4, atol=2e-5)
    loss_sum.backward()
    for k in [k for k in d.files if k.startswith("train/grad/")]:
        p = st[k[len("train/grad/"):]]
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        ok, msg = digest_close(d[k], g.numpy(), rtol=2e-3, atol=1e-7)
        assert ok, f"{k}: {msg}"
    for k in [k for k in d.files if k.startswith("train/after/")]:
        ok, msg = digest_close(d[k], st[k[len("train/after/"):]].detach().numpy(), rtol=1e-4, atol=1e-7)
        assert ok, f"{k}: {msg}"


def test_dataset_weight_variant(golden_dir):
    d, meta = _load(golden_dir, "model_full.npz")
    shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
    crit, _ = _criterions(meta, golden_dir)
    image, ids = make_inputs(meta["B"], seed=meta["input_seed"])
    st = R.state_from_numpy(make_state(shapes, meta["state_seed"]), requires_grad=False)
    with torch.no_grad():
        out, _ = R.network_forward(st, torch.from_numpy(image), torch.from_numpy(ids), meta["config"], True)
        loss_sum, by_name = R.compute_loss(out, _batches(meta, True), 150, crit)
    np.testing.assert_allclose(loss_sum.item(), d["train_dw/loss_sum"], rtol=1e-4)
    for n, (_, w) in by_name.items():
        np.testing.assert_allclose(w.numpy(), d[f"train_dw/loss/{n}/weights"], rtol=1e-6)


def test_validation_criterion(golden_dir):
    d, meta = _load(golden_dir, "model_full.npz")
    sd = make_state({k: tuple(v) for k, v in meta["shapes"].items()}, meta["state_seed"])
    sd.update({k[len("calib/"):]: d[k] for k in d.files if k.startswith("calib/")})
    st = R.state_from_numpy(sd, requires_grad=False)
    _, test_crit = _criterions(meta, golden_dir)
    image, _ = make_inputs(meta["B"], seed=meta["input_seed"])
    b0 = _batches(meta)[0]
    with torch.no_grad():
        out, _ = R.network_forward(st, torch.from_numpy(image[: meta["split"]]), None, meta["config"], False)
        terms = [(n, f(out, b0), w(3) if callable(w) else w) for n, f, w in test_crit["POSE_WITH_LANDMARKS"]]
    assert [t[0] for t in terms] == json.loads(str(d["val/names"]))
    val_loss = sum((v * w).sum() for _, v, w in terms)
    np.testing.assert_allclose(val_loss.item(), d["val/val_loss"], rtol=2e-4)


@pytest.mark.parametrize("cfg", ["full", "default"])
def test_clip_adam_three_steps(cfg, golden_dir):
    d, meta = _load(golden_dir, f"model_{cfg}.npz")
    o = np.load(os.path.join(golden_dir, f"optim_{cfg}.npz"))
    ometa = json.loads(str(o["meta"]))
    shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
    fl = meta["flags"]
    gmm = R.ShapeGmm(os.path.join(golden_dir, "shapeparams_gmm.npz"))
    crit, _ = R.setup_losses(with_pointhead=fl["with_pointhead"], with_nll_loss=fl["with_nll_loss"],
                             rampup_nll_losses=fl["rampup_nll_losses"], epochs=ometa["epochs"], gmm=gmm)
    st = R.state_from_numpy(make_state(shapes, meta["state_seed"]))
    opt = R.ClipAdam(st, lr=ometa["lr"], epochs=ometa["epochs"])
    assert [len(g) for g, _ in opt.groups] == ometa["group_sizes"][:2]
    image, ids = make_inputs(meta["B"], seed=meta["input_seed"])
    for step in range(ometa["steps"]):
        opt.zero_grad()
        out, _ = R.network_forward(st, torch.from_numpy(image), torch.from_numpy(ids), meta["config"], True)
        loss_sum, _ = R.compute_loss(out, _batches(meta), step, crit)
        loss_sum.backward()
        # Step 0 is reproducible to rounding.  Later steps are not: Adam's first update moves every
        # weight by lr*sign(g), and where g is at rounding level the sign differs between two fp32
        # implementations (B=8, BN over 200 samples amplifies it), so steps 1-2 are a plausibility check
        # only; the sharp optimiser test is test_clip_adam_fixed_gradients below.
        tol = 1e-4 if step == 0 else 5e-2
        np.testing.assert_allclose(loss_sum.item(), o[f"step{step}/loss_sum"], rtol=tol)
        np.testing.assert_allclose(np.array(opt.lrs()), o[f"step{step}/lrs"][:2], rtol=1e-6)
        gn = opt.step()
        np.testing.assert_allclose(gn.item(), o[f"step{step}/grad_norm"], rtol=max(tol, 2e-3))
        opt.end_epoch()


@pytest.mark.parametrize("cfg", ["full", "default"])
def test_clip_adam_fixed_gradients(cfg, golden_dir):
    """clip(1.0) + 2-group Adam + per-epoch LR schedule on FIXED synthetic gradients: reproducible to
    rounding, so parameters are compared tightly."""
    d, meta = _load(golden_dir, f"model_{cfg}.npz")
    o = np.load(os.path.join(golden_dir, f"optim_{cfg}.npz"))
    shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
    st = R.state_from_numpy(make_state(shapes, meta["state_seed"]))
    opt = R.ClipAdam(st, lr=1.0e-3, epochs=20)
    pshapes = {k: tuple(v.shape) for k, v in st.items() if not R.is_buffer(k)}
    for step, gscale in enumerate((1.0e-3, 1.0e-4, 1.0e-2)):
        g = make_grads(pshapes, seed=200 + step, scale=gscale)
        for k in pshapes:
            st[k].grad = torch.from_numpy(g[k].copy())
        gn = opt.step()
        np.testing.assert_allclose(gn.item(), o[f"fixed/step{step}/grad_norm"], rtol=1e-5)
        opt.end_epoch()
    for k in [k for k in o.files if k.startswith("fixed/final/")]:
        ok, msg = digest_close(o[k], st[k[len("fixed/final/"):]].detach().numpy(), rtol=1e-5, atol=2e-7)
        assert ok, f"{k}: {msg}"


def test_lr_schedule_tables(golden_dir):
    s = np.load(os.path.join(golden_dir, "schedule.npz"))
    for E in (200, 1500):
        mine = np.array([R.lr_factor(e, E) for e in range(E)])
        np.testing.assert_allclose(mine, s[f"E{E}"], rtol=1e-12)


def test_swa_average(golden_dir):
    s = np.load(os.path.join(golden_dir, "swa.npz"))
    keys = [k for k in s.files if k != "n_averaged"]
    shapes = {k: tuple(s[k].shape) for k in keys}
    avg = {k: torch.zeros(shapes[k], dtype=torch.from_numpy(s[k]).dtype) for k in keys}
    for i in range(3):
        sd = make_state({("bn." + k if k.startswith("1.") else k): v for k, v in shapes.items()}, seed=100 + i)
        sd = {(k[3:] if k.startswith("bn.") else k): torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        sd["1.num_batches_tracked"] = torch.tensor(i + 1, dtype=torch.int64)
        R.swa_update(avg, sd, i)
    assert int(s["n_averaged"]) == 3
    for k in keys:
        np.testing.assert_allclose(avg[k].numpy(), s[k], rtol=1e-6, atol=1e-7, err_msg=k)
