"""Parity of the whole training step at the sizes the metric is quoted on (BASELINE configs 2 and 4/1-GPU share:
B = 256 and B = 512 per GPU): the HIP step against the CPU oracle on identical weights and inputs.

At these sizes every kernel runs its multi-round grids (2 163 200-row GEMMs, > 1024 partial rows folded by
bn_fold_rows_k, XCD tile remapping over thousands of tiles, persistent depthwise workgroups looping over dozens of
tiles) - none of which the B <= 53 tests reach.  Criteria (north_star): per-step losses and loss_sum within 1e-3,
features within 1e-4 relative; every parameter gradient as close to the fp64 oracle as the reference's own fp32 CPU
arithmetic is (3 x its error + 1e-5), with NO trimming of outliers - at this batch a ReLU decision that differs
between two fp32 evaluations moves a gradient by 1/B of what it does at B = 8."""
import gc
import itertools
import os

import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.synth import make_inputs, make_state
from util import GOLDEN, build_net, load_golden, make_batches, script_args, train_script

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _oracle(meta, shapes, image, ids, epoch, dtype, want_grads):
    from test_oracle_golden import _batches, _criterions

    crit, _ = _criterions(meta, GOLDEN)
    st = {}
    for k, v in make_state(shapes, meta["state_seed"]).items():
        t = torch.from_numpy(np.array(v))
        t = t.to(dtype) if t.is_floating_point() else t
        st[k] = t.requires_grad_(True) if (want_grads and not R.is_buffer(k)) else t
    batches = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()} for b in _batches(meta)]
    with torch.set_grad_enabled(want_grads):
        out, feat = R.network_forward(st, torch.from_numpy(image).to(dtype), torch.from_numpy(ids), meta["config"], True)
        loss, by_name = R.compute_loss(out, batches, epoch, crit)
    grads = None
    if want_grads:
        loss.backward()
        grads = {k: v.grad for k, v in st.items() if not R.is_buffer(k)}
    res = dict(loss=float(loss.item()), by_name={k: v[0].detach().clone() for k, v in by_name.items()}, feat=feat.detach().clone(), grads=grads,
               running={k: v.detach().clone() for k, v in st.items() if k.endswith("running_var") or k.endswith("running_mean")})
    del st, out, feat, loss, by_name
    gc.collect()
    return res


# (golden config, per-GPU batch, epoch): "full" = uncertainty heads + NLL losses whose weights ramp with the epoch
@pytest.mark.parametrize("cfg,B,epoch", [("default", 512, 0), ("full", 512, 150), ("default", 256, 150), ("full", 256, 0)])
def test_step_at_benchmark_size_matches_oracle(cfg, B, epoch):
    import trackertraincode.train as train

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    _, meta = load_golden(f"model_{cfg}.npz")
    meta = dict(meta, B=B, split=(B * 5) // 8)  # two Tags: POSE_WITH_LANDMARKS + ONLY_POSE
    shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
    image, ids = make_inputs(B, seed=meta["input_seed"])
    S = train_script()

    net = build_net(meta, DEV).train()
    crit, _ = S.setup_losses(script_args(meta["flags"]), net)
    feats = []
    orig = net.convnet.forward_features
    net.convnet.forward_features = lambda x: feats.append(orig(x)) or feats[-1]
    batches = make_batches(meta, DEV)
    inputs = torch.concat([b["image"] for b in batches], dim=0)
    ids_d = torch.concat([b["coord_convention_id"] for b in batches], dim=0)
    preds = net(inputs, ids_d)
    loss_sum, all_lossvals = train.default_compute_loss(preds, batches, epoch, crit)
    by_name = train.concatenated_lossvals_by_name(itertools.chain.from_iterable(all_lossvals))
    loss_sum.backward()
    torch.cuda.synchronize()
    hip_grads = {k: (None if p.grad is None else p.grad.detach().cpu()) for k, p in net.named_parameters()}
    hip_feat = feats[0].detach().cpu()
    hip_loss = loss_sum.item()
    hip_vals = {k: v[0].detach().cpu() for k, v in by_name.items()}
    hip_state = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    del net, preds, loss_sum, by_name, all_lossvals, feats
    torch.cuda.empty_cache()

    o32 = _oracle(meta, shapes, image, ids, epoch, torch.float32, want_grads=True)
    # ---- losses and features against the fp32 oracle (the reference's arithmetic)
    assert abs(hip_loss - o32["loss"]) < 1e-3, (hip_loss, o32["loss"])
    assert list(hip_vals.keys()) == list(o32["by_name"].keys())
    for n, v in o32["by_name"].items():
        np.testing.assert_allclose(hip_vals[n].numpy(), v.numpy(), rtol=1e-3, atol=1e-3, err_msg=n)
    e_feat = _rel(hip_feat, o32["feat"])
    assert e_feat < 1e-4, e_feat
    # BatchNorm running statistics after one step (momentum 0.1, unbiased variance)
    for k, v in o32["running"].items():
        np.testing.assert_allclose(hip_state[k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)

    # ---- gradients: as close to fp64 as the fp32 CPU path is
    o64 = _oracle(meta, shapes, image, ids, epoch, torch.float64, want_grads=True)
    assert abs(hip_loss - o64["loss"]) < 1e-3
    bad, worst = [], (0.0, "")
    for k, g in hip_grads.items():
        g64 = o64["grads"][k]
        if g64 is None:  # parameter no active loss depends on
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        e_hip, e_cpu = _rel(g, g64), _rel(o32["grads"][k], g64)
        if e_hip > worst[0]:
            worst = (e_hip, k)
        if e_hip > 3 * e_cpu + 1e-5:
            bad.append((k, f"hip {e_hip:.2e}", f"cpu32 {e_cpu:.2e}"))
    print(f"cfg={cfg} B={B} epoch={epoch}: loss {hip_loss:.6f} (oracle {o32['loss']:.6f}), features rel {e_feat:.1e}, worst gradient rel {worst[0]:.1e} ({worst[1]})")
    assert not bad, bad[:8]
