"""Worker of tests/test_trajectory_gpu.py (its own process: TTK_DETERMINISTIC is read when the package is imported).

K optimiser steps of the REAL loop - zero_grad -> forward -> multi-task loss -> backward -> global-norm clip -> Adam ->
the next forward with the updated weights and running statistics - on the HIP path, and the same K steps on the CPU
oracle (oracle/refmodel.py: network_forward + compute_loss + ClipAdam) from identical weights and inputs.
Reference loop: trackertraincode/train.py:372-439, scripts/train_poseestimator.py:147-167,442-454.

Prints one line "RESULT <json>": per step |loss_sum difference| and the largest per-sample loss difference, after the
last step the BatchNorm running statistics' and the parameters' distance.
With `f64` the walk is also made by the oracle in float64, and the distances HIP <-> fp64 and fp32 oracle <-> fp64 are reported:
how far two fp32 implementations of the same loop part from one another is the yardstick for the HIP walk's distance.
usage: _trajectory_worker.py <repo> <cfg> <B> <steps> <loss epoch> <epochs of the lr schedule> <lr epoch> <f64|nof64>
"""
import json
import os
import sys

repo, cfg, B, K, epoch = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
LR_EPOCHS, LR_EPOCH, with_f64 = int(sys.argv[6]), int(sys.argv[7]), sys.argv[8] == "f64"
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

# schedule position of both optimisers (ExponentialUpThenSteps over LR_EPOCHS epochs, at epoch LR_EPOCH): (200, 0) = the first epoch
# of the training script's default run, lr 1.26e-5; (20, 5) = past the warm-up, the full lr 1e-3
LR0 = 1.0e-3 * R.lr_factor(LR_EPOCH, LR_EPOCHS)
_, meta = load_golden(f"model_{cfg}.npz")
meta = dict(meta, B=B, split=(B * 5) // 8)
S = train_script()

# ---- HIP trajectory
with gpu_section():
    net = build_net(meta, "cuda").train()
    crit, _ = S.setup_losses(script_args(meta["flags"]), net)
    opt, sch = S.create_optimizer(net, script_args(meta["flags"], epochs=LR_EPOCHS))
    import warnings  # noqa: E402

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(LR_EPOCH):
            sch.step()
    assert abs(opt.param_groups[0]["lr"] - LR0) < 1e-12 * max(1.0, LR0), (opt.param_groups[0]["lr"], LR0)
    batches = make_batches(meta, "cuda")
    hip_loss, hip_vals, hip_norm = [], [], []
    for it in range(K):
        opt.zero_grad(set_to_none=True)
        out = train.training_step(net, batches, epoch, crit)
        out["loss"].backward()
        opt.step()
        hip_loss.append(out["loss"].item())
        hip_vals.append({k: v.detach().cpu().double() for k, v in out["mt_losses"].items()})
        hip_norm.append(float(opt.last_grad_norm.item()))
    torch.cuda.synchronize()
    hip_state = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    del net, opt, out
    torch.cuda.empty_cache()

# ---- oracle trajectories (CPU): fp32 = the reference's arithmetic; fp64 (optional) = the yardstick for how far two fp32 walks may part
shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
ocrit, _ = _criterions(meta, GOLDEN)
image, ids = make_inputs(B, seed=meta["input_seed"])


def oracle_walk(dtype):
    st = {}
    for k, v in make_state(shapes, meta["state_seed"]).items():
        t = torch.from_numpy(np.array(v))
        t = t.to(dtype) if t.is_floating_point() else t
        st[k] = t.requires_grad_(True) if not R.is_buffer(k) else t
    init = {k: v.detach().clone().double() for k, v in st.items()}
    oopt = R.ClipAdam(st, lr=1.0e-3, epochs=LR_EPOCHS)
    oopt.epoch = LR_EPOCH
    assert abs(oopt.lrs()[0] - LR0) < 1e-12 * max(1.0, LR0), (oopt.lrs(), LR0)
    x, ids_t = torch.from_numpy(image).to(dtype), torch.from_numpy(ids)
    obatches = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()} for b in _batches(meta)]
    losses, vals, norms = [], [], []
    for it in range(K):
        oopt.zero_grad()
        o, _ = R.network_forward(st, x, ids_t, meta["config"], True)
        loss, by_name = R.compute_loss(o, obatches, epoch, ocrit)
        loss.backward()
        norms.append(float(oopt.step()))
        losses.append(float(loss.item()))
        vals.append({n: v[0].detach().double().clone() for n, v in by_name.items()})
        del o, loss, by_name
    return losses, vals, norms, {k: v.detach().double().clone() for k, v in st.items()}, init


def sample_dist(a, b):  # largest per-sample loss difference of one step
    assert list(a.keys()) == list(b.keys())
    return max(float((a[n] - b[n]).abs().max()) for n in a)


def state_dist(a, b, init):
    run_rel, par_abs, par_rel, moved, worst_run, worst_par = 0.0, 0.0, 0.0, 0.0, "", ""
    for k, v in b.items():
        x, y = a[k], v
        if k.endswith("num_batches_tracked"):
            assert int(x) == int(y) == K, (k, int(x), int(y))
        elif "running_" in k:
            e = float(((x - y).abs() / (y.abs() + 1e-2 * float(y.abs().max()) + 1e-30)).max())
            if e > run_rel:
                run_rel, worst_run = e, k
        elif not R.is_buffer(k):
            d = float((x - y).abs().max())
            if d > par_abs:
                par_abs, worst_par = d, k
            step_len = float((y - init[k]).norm())
            moved = max(moved, float((y - init[k]).abs().max()))
            if step_len > 0:
                par_rel = max(par_rel, float((x - y).norm()) / step_len)
    return dict(running_rel=run_rel, worst_running=worst_run, param_abs=par_abs, worst_param=worst_par, param_rel_to_path=par_rel, largest_param_move=moved)


l32, v32, n32, s32, init = oracle_walk(torch.float32)
res = dict(cfg=cfg, B=B, steps=K, lr=LR0, loss_hip=hip_loss, loss_oracle=l32, gnorm_hip=hip_norm, gnorm_oracle=n32,
           dloss=[abs(a - b) for a, b in zip(hip_loss, l32)], dsample=[sample_dist(hip_vals[i], v32[i]) for i in range(K)])
res.update(state_dist(hip_state, s32, init))
if with_f64:
    l64, v64, n64, s64, _ = oracle_walk(torch.float64)
    res.update(dloss_hip_64=[abs(a - b) for a, b in zip(hip_loss, l64)], dloss_cpu32_64=[abs(a - b) for a, b in zip(l32, l64)],
               dsample_hip_64=[sample_dist(hip_vals[i], v64[i]) for i in range(K)], dsample_cpu32_64=[sample_dist(v32[i], v64[i]) for i in range(K)])
    res["state_hip_64"] = state_dist(hip_state, s64, init)
    res["state_cpu32_64"] = state_dist(s32, s64, init)
print("RESULT " + json.dumps(res))
