"""Worker of tests/test_teacher_forced_gpu.py (its own process: TTK_DETERMINISTIC is read when the library is loaded).

TEACHER-FORCED walk: the fp32 CPU oracle (oracle/refmodel.py = the reference's arithmetic) makes K optimiser steps of the real loop
(zero_grad -> forward -> multi-task loss -> backward -> global-norm clip -> Adam; reference trackertraincode/train.py:372-439,
scripts/train_poseestimator.py:147-167,442-454) and records, for every step t, its complete state BEFORE the step - parameters,
BatchNorm buffers, Adam moments, step counts - and what the step produced.  For every t the HIP model and the fused ClipAdam are
loaded with the oracle's state at t, make ONE step, and are compared with the oracle's step t -> t + 1.  Unlike a free walk
(tests/test_trajectory_gpu.py), whose two fp32 implementations part chaotically after the first Adam update, every comparison here
starts from IDENTICAL state, so optimiser-state and running-statistic evolution are pinned at every t with single-step tolerances.

Prints "RESULT <json>": per step |loss_sum difference|, largest per-sample loss difference, running-statistics distance, Adam moment
distances, gradient-norm difference, parameter distance after the update.
usage: _teacher_forced_worker.py <repo> <cfg> <B> <steps> <loss epoch>
"""
import json
import os
import sys

repo, cfg, B, K, epoch = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
for p_ in (repo, repo + "/neuralnet-tracker-traincode_amd", repo + "/tests"):
    sys.path.insert(0, p_)
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(min(os.cpu_count() or 1, 32))
from oracle import refmodel as R  # noqa: E402
from oracle.synth import make_inputs, make_state  # noqa: E402
from test_oracle_golden import _batches, _criterions  # noqa: E402
from util import GOLDEN, build_net, gpu_section, load_golden, make_batches, script_args, train_script  # noqa: E402
import trackertraincode.train as train  # noqa: E402

LR_EPOCHS, LR_EPOCH = 20, 5  # past the warm-up of ExponentialUpThenSteps: the full learning rate 1e-3
LR0 = 1.0e-3 * R.lr_factor(LR_EPOCH, LR_EPOCHS)
_, meta = load_golden(f"model_{cfg}.npz")
meta = dict(meta, B=B, split=(B * 5) // 8)
S = train_script()
shapes = {k: tuple(v) for k, v in meta["shapes"].items()}

# ---- the oracle's walk, with a snapshot in front of every step
ocrit, _ = _criterions(meta, GOLDEN)
image, ids = make_inputs(B, seed=meta["input_seed"])
st = {}
for k, v in make_state(shapes, meta["state_seed"]).items():
    t = torch.from_numpy(np.array(v))
    st[k] = t.requires_grad_(True) if not R.is_buffer(k) else t
oopt = R.ClipAdam(st, lr=1.0e-3, epochs=LR_EPOCHS)
oopt.epoch = LR_EPOCH
x, ids_t = torch.from_numpy(image), torch.from_numpy(ids)
obatches = _batches(meta)
snaps = []
for it in range(K):
    snap = dict(state={k: v.detach().clone() for k, v in st.items()}, m={k: v.clone() for k, v in oopt.m.items()}, v={k: v.clone() for k, v in oopt.v.items()}, t=oopt.t)
    oopt.zero_grad()
    o, _ = R.network_forward(st, x, ids_t, meta["config"], True)
    loss, by_name = R.compute_loss(o, obatches, epoch, ocrit)
    loss.backward()
    snap["gnorm"] = float(oopt.step())
    snap["loss"] = float(loss.item())
    snap["vals"] = {n: v[0].detach().double().clone() for n, v in by_name.items()}
    snap["after"] = {k: v.detach().clone() for k, v in st.items()}
    snap["m_after"] = {k: v.clone() for k, v in oopt.m.items()}
    snap["v_after"] = {k: v.clone() for k, v in oopt.v.items()}
    snaps.append(snap)
    del o, loss, by_name


def f64_moments(snap):
    """The same step from the same state in float64: the yardstick for the moments (the fp32 oracle's own distance to it)."""
    st64 = {k: (v.double().requires_grad_(True) if not R.is_buffer(k) else (v.double() if v.is_floating_point() else v.clone())) for k, v in snap["state"].items()}
    o64 = R.ClipAdam(st64, lr=1.0e-3, epochs=LR_EPOCHS)
    o64.epoch, o64.t = LR_EPOCH, snap["t"]
    o64.m = {k: v.double().clone() for k, v in snap["m"].items()}
    o64.v = {k: v.double().clone() for k, v in snap["v"].items()}
    b64 = [{k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()} for b in obatches]
    o, _ = R.network_forward(st64, x.double(), ids_t, meta["config"], True)
    loss, _ = R.compute_loss(o, b64, epoch, ocrit)
    loss.backward()
    o64.step()
    return {k: v.clone() for k, v in o64.m.items()}, {k: v.clone() for k, v in o64.v.items()}


YARD_T = K - 1
m64, v64 = f64_moments(snaps[YARD_T])

# ---- HIP: one step from every snapshot (the workers' GPU parts take turns: util.gpu_section; released when the process ends)
_turn = gpu_section()
_turn.__enter__()
net = build_net(meta, "cuda").train()
crit, _ = S.setup_losses(script_args(meta["flags"]), net)
opt, sch = S.create_optimizer(net, script_args(meta["flags"], epochs=LR_EPOCHS))
import warnings  # noqa: E402

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(LR_EPOCH):
        sch.step()
assert abs(opt.param_groups[0]["lr"] - LR0) < 1e-12, (opt.param_groups[0]["lr"], LR0)
batches = make_batches(meta, "cuda")
names = {id(p): n for n, p in net.named_parameters()}
res = dict(cfg=cfg, B=B, steps=K, lr=LR0, dloss=[], dsample=[], running_rel=[], worst_running=[], dgnorm_rel=[], m_rel=[], v_rel=[], m_abs=[], worst_m=[],
           param_abs=[], param_over_lr=[], loss_oracle=[], loss_hip=[],
           m_l2_worst=[], m_l2_median=[], v_l2_worst=[], v_l2_median=[])
for it, snap in enumerate(snaps):
    net.load_state_dict({k: v for k, v in snap["state"].items()}, strict=True)
    base = opt.state_dict()
    state, idx = {}, 0
    for g in opt.param_groups:
        for p in g["params"]:
            n = names[id(p)]
            if n in snap["m"]:
                state[idx] = {"step": torch.tensor(float(snap["t"])), "exp_avg": snap["m"][n].clone(), "exp_avg_sq": snap["v"][n].clone()}
            idx += 1
    opt.load_state_dict({"state": state if snap["t"] > 0 else {}, "param_groups": base["param_groups"]})
    opt.zero_grad(set_to_none=True)
    out = train.training_step(net, batches, epoch, crit)
    out["loss"].backward()
    opt.step()
    torch.cuda.synchronize()
    res["loss_hip"].append(out["loss"].item())
    res["loss_oracle"].append(snap["loss"])
    res["dloss"].append(abs(out["loss"].item() - snap["loss"]))
    vals = {k: v.detach().cpu().double() for k, v in out["mt_losses"].items()}
    assert list(vals.keys()) == list(snap["vals"].keys())
    res["dsample"].append(max(float((vals[n] - snap["vals"][n]).abs().max()) for n in vals))
    res["dgnorm_rel"].append(abs(float(opt.last_grad_norm.item()) - snap["gnorm"]) / snap["gnorm"])
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    run, worst = 0.0, ""
    for k, y in snap["after"].items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(y), (k, int(sd[k]), int(y))
        elif "running_" in k:
            e = float(((sd[k].double() - y.double()).abs() / (y.double().abs() + 1e-2 * float(y.abs().max()) + 1e-30)).max())
            if e > run:
                run, worst = e, k
    res["running_rel"].append(run)
    res["worst_running"].append(worst)
    # Adam moments: relative to the largest element of the oracle's moment tensor (a per-tensor scale: single elements of a moment can be zero)
    m_rel = v_rel = m_abs = 0.0
    worst_m = ""
    par_abs = 0.0
    m_l2, v_l2 = [], []  # per tensor: ||hip - oracle||_2 / ||oracle||_2
    for g in opt.param_groups:
        for p in g["params"]:
            n = names[id(p)]
            if n not in snap["m_after"] or p not in opt.state:
                continue
            s_ = opt.state[p]
            if p.grad is None:  # a parameter no active loss depends on: neither Adam advances it (torch.optim.Adam skips it, so does the oracle)
                assert float((p.detach().cpu() - snap["after"][n]).abs().max()) == 0.0, n
                continue
            assert int(float(s_["step"])) == snap["t"] + 1, (n, float(s_["step"]), snap["t"])
            mo, vo = snap["m_after"][n].double(), snap["v_after"][n].double()
            dm = float((s_["exp_avg"].cpu().double() - mo).abs().max())
            e = dm / (float(mo.abs().max()) + 1e-30)
            if e > m_rel:
                m_rel, worst_m = e, n
            m_abs = max(m_abs, dm)
            m_l2.append(float((s_["exp_avg"].cpu().double() - mo).norm() / (mo.norm() + 1e-30)))
            v_l2.append(float((s_["exp_avg_sq"].cpu().double() - vo).norm() / (vo.norm() + 1e-30)))
            v_rel = max(v_rel, float((s_["exp_avg_sq"].cpu().double() - vo).abs().max()) / (float(vo.abs().max()) + 1e-30))
            par_abs = max(par_abs, float((p.detach().cpu().double() - snap["after"][n].double()).abs().max()))
    res["m_rel"].append(m_rel)
    res["v_rel"].append(v_rel)
    res["m_abs"].append(m_abs)
    if it == YARD_T:  # per tensor: HIP vs float64 against fp32 oracle vs float64
        ratio, worst_ratio = 0.0, ""
        for g in opt.param_groups:
            for p in g["params"]:
                n = names[id(p)]
                if n not in m64 or p not in opt.state:
                    continue
                for kind, hip_t, or_t, ref in (("exp_avg", opt.state[p]["exp_avg"], snap["m_after"][n], m64[n]), ("exp_avg_sq", opt.state[p]["exp_avg_sq"], snap["v_after"][n], v64[n])):
                    e_hip = float((hip_t.cpu().double() - ref).norm() / (ref.norm() + 1e-30))
                    e_cpu = float((or_t.double() - ref).norm() / (ref.norm() + 1e-30))
                    r_ = e_hip / (3 * e_cpu + 1e-5)
                    if r_ > ratio:
                        ratio, worst_ratio = r_, f"{n} {kind}: hip {e_hip:.2e} cpu32 {e_cpu:.2e}"
        res["yard_ratio"], res["yard_worst"] = ratio, worst_ratio
    res["m_l2_worst"].append(max(m_l2)); res["m_l2_median"].append(float(np.median(m_l2)))
    res["v_l2_worst"].append(max(v_l2)); res["v_l2_median"].append(float(np.median(v_l2)))
    res["worst_m"].append(worst_m)
    res["param_abs"].append(par_abs)
    res["param_over_lr"].append(par_abs / LR0)
    del out
print("RESULT " + json.dumps(res))
