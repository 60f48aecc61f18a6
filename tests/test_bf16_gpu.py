"""bf16 activation storage (`--precision bf16`, BASELINE config 5): raw conv outputs, materialised block inputs and their
gradients cross HBM as bfloat16 (round to nearest even); BatchNorm statistics are taken from the ROUNDED values, all
arithmetic, weights, statistics and weight gradients stay fp32.  The reference has no such mode (it runs fp32 only,
scripts/train_poseestimator.py:442-454): parity is stated against the fp32 path / oracle with the MEASURED tolerance
asserted here, and the bench reports it as a separate line (dtype bf16), never as the headline."""
import itertools

import numpy as np
import pytest
import torch

from util import build_net, load_golden, make_batches, script_args, train_script

pytestmark = pytest.mark.gpu
DEV = "cuda"
BN_SCALE, BN_BETA, BN_MEAN, BN_RSTD, BN_GA, BN_GB, BN_GMEAN, BN_AUX = range(8)


def _bn_block(C, rng):
    bn = np.zeros((8, C), np.float32)
    bn[BN_SCALE], bn[BN_BETA], bn[BN_MEAN] = rng.uniform(0.5, 1.5, C), rng.normal(0, 0.2, C), rng.normal(0, 0.3, C)
    bn[BN_RSTD], bn[BN_GA], bn[BN_GB], bn[BN_GMEAN] = rng.uniform(0.5, 2.0, C), rng.uniform(0.5, 1.5, C), rng.normal(0, 0.2, C), rng.normal(0, 0.05, C)
    return bn


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("flag", [1, 3])  # TTK_STORE_ACT_BF16 (gradients fp32) / | TTK_STORE_GRAD_BF16 (gradients bf16 too)
@pytest.mark.parametrize("M,Cin,Cout", [(648, 512, 512), (4100, 256, 256), (777, 256, 128), (1234, 32, 64), (5000, 128, 128), (900, 64, 128)])
def test_pointwise_kernels_with_bf16_storage(M, Cin, Cout, flag):
    """Inputs are bf16 tensors; the fp64 reference is formed from exactly those values, so the only differences left are
    the rounding of the stored outputs (2^-9 relative per element) - and the partial sums must be those of the STORED
    outputs to fp32 accuracy."""
    import trackertraincode._hip as H
    L, p = H.lib(), H.ptr
    rng = np.random.default_rng(M + Cin + Cout)
    bf = lambda a: H.to_blocks(torch.from_numpy(a.astype(np.float32)).to(DEV).to(torch.bfloat16))  # activations: channel blocks (include/ttk.h)
    gdt = torch.bfloat16 if flag == 3 else torch.float32  # storage of the gradient tensors
    gr = lambda a: H.to_blocks(torch.from_numpy(a.astype(np.float32)).to(DEV).to(gdt))
    f64 = lambda t: H.from_blocks(t).float().cpu().numpy().astype(np.float64)
    ydw = bf(rng.normal(0, 1, (M, Cin)))
    w = (rng.normal(0, 1, (Cout, Cin)) * np.sqrt(2.0 / Cout)).astype(np.float32)
    bn_dw, bn_pw = _bn_block(Cin, rng), _bn_block(Cout, rng)
    a64 = np.maximum(bn_dw[BN_SCALE].astype(np.float64) * (f64(ydw) - bn_dw[BN_MEAN]) + bn_dw[BN_BETA], 0)
    bn_dw[BN_AUX, 0] = np.abs(a64).max() * 3
    y64 = a64 @ w.astype(np.float64).T
    d_w, d_bn = torch.from_numpy(w).to(DEV), torch.from_numpy(bn_dw).to(DEV)
    rows = L.partial_rows_gemm(M, Cin, Cout)
    y = torch.empty(M, Cout, device=DEV, dtype=torch.bfloat16)
    part = torch.full((rows, 2, Cout), float("nan"), device=DEV)
    wq = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device=DEV)
    L.call("ttk_pwconv1x1_fwd", p(ydw), p(d_bn), p(d_w), p(y), p(part), None, M, Cin, Cout, p(wq), flag)
    torch.cuda.synchronize()
    assert _rel(f64(y), y64) < 3e-3  # bf16 rounding of the outputs: 2^-9 per element
    ps, ys = part.cpu().numpy().astype(np.float64), f64(y)
    np.testing.assert_allclose(ps[:, 0].sum(0), ys.sum(0), rtol=0, atol=2e-5 * np.abs(ys).sum(0).max())  # sums of what is STORED
    np.testing.assert_allclose(ps[:, 1].sum(0), (ys ** 2).sum(0), rtol=2e-5)
    # ---- data gradient
    g = gr(rng.normal(0, 1, (M, Cout)))
    dy64 = bn_pw[BN_GA].astype(np.float64) * (f64(g) - bn_pw[BN_GMEAN]) + bn_pw[BN_GB].astype(np.float64) * (ys - bn_pw[BN_MEAN])
    bn_pw[BN_AUX, 1] = np.abs(dy64).max() * 3
    pre = bn_dw[BN_SCALE].astype(np.float64) * (f64(ydw) - bn_dw[BN_MEAN]) + bn_dw[BN_BETA]
    safe = np.abs(pre) > 1e-4
    gd64 = (dy64 @ w.astype(np.float64)) * (pre > 0)
    wt = torch.from_numpy(np.ascontiguousarray(w.T)).to(DEV)
    g_dw = torch.empty(M, Cin, device=DEV, dtype=gdt)
    part2 = torch.full((L.partial_rows_gemm(M, Cout, Cin, True), 2, Cin), float("nan"), device=DEV)
    d_bnpw = torch.from_numpy(bn_pw).to(DEV)
    L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(d_bnpw), p(wt), p(ydw), p(d_bn), p(g_dw), p(part2), M, Cin, Cout, p(wq), flag)
    torch.cuda.synchronize()
    out = f64(g_dw)
    assert _rel(out * safe, gd64 * safe) < (3e-3 if flag == 3 else 2e-6)  # fp32 gradient storage: full accuracy
    ps = part2.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(ps[:, 0].sum(0), out.sum(0), rtol=0, atol=2e-5 * np.abs(out).sum(0).max())
    np.testing.assert_allclose(ps[:, 1].sum(0), (out * (f64(ydw) - bn_dw[BN_MEAN])).sum(0), rtol=0, atol=2e-5 * np.abs(out * (f64(ydw) - bn_dw[BN_MEAN])).sum(0).max())
    # ---- weight gradient (fp32 output): full fp32 accuracy on the bf16-valued operands
    dw64 = dy64.T @ a64
    dW = torch.zeros(Cout, Cin, device=DEV)
    L.call("ttk_pwconv1x1_bwd_weight", p(g), p(y), p(d_bnpw), p(ydw), p(d_bn), p(dW), None, M, Cin, Cout, flag)
    torch.cuda.synchronize()
    assert _rel(dW.cpu().numpy(), dw64) < 2e-6


def _step(meta, epoch, mode):
    import trackertraincode.backbones.mobilenet_v1 as MB
    import trackertraincode.train as train

    S = train_script()
    MB.set_activation_dtype(mode)
    try:
        net = build_net(meta, DEV).train()
        crit, _ = S.setup_losses(script_args(meta["flags"]), net)
        feats = []
        orig = net.convnet.forward_features
        net.convnet.forward_features = lambda x: feats.append(orig(x)) or feats[-1]
        batches = make_batches(meta, DEV)
        out = train.training_step(net, batches, epoch, crit)
        out["loss"].backward()
        torch.cuda.synchronize()
        return dict(loss=out["loss"].item(), feat=feats[0].detach().float().cpu(), mt={k: v.cpu() for k, v in out["mt_losses"].items()},
                    grads={k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None},
                    state={k: v.detach().cpu() for k, v in net.state_dict().items()})
    finally:
        MB.set_activation_dtype("fp32")


@pytest.mark.parametrize("cfg,B,mode", [("default", 96, "bf16"), ("full", 256, "bf16"), ("default", 96, "bf16-all")])
def test_bf16_step_tracks_the_fp32_step(cfg, B, mode):
    """Whole training step with bf16 storage against the same step in fp32 (which the other tests hold to the reference).
    The assertions are the MEASURED deviations on MI355X with headroom: loss 6e-4 relative, pooled features 2.6e-2,
    per-sample losses 7e-2 of their scale, running statistics 1e-3, total gradient norm a few per cent, heads' gradients
    cosine 1.000.  Single-step parameter gradients of the backbone are far more sensitive: they are sums over 10^5-10^6
    pixels that cancel to a small remainder and depend on ReLU decisions - even fp32 against fp64 moves the worst tensor by
    7e-3 (tests/test_fullsize_gpu.py), an amplification of ~10^5 of the rounding, so 2^-9 perturbations of the stored
    activations leave per-tensor cosines of 0.69-0.97 against the fp32 step (worst tensor of each block, smallest at the
    input end; the same with the gradients stored in fp32 or in bf16: the activations' rounding dominates).  Reported, and
    bounded loosely, here; whether training converges the same is a question for a long run with data, not for a step."""
    _, meta = load_golden(f"model_{cfg}.npz")
    meta = dict(meta, B=B, split=(B * 5) // 8)
    a, b = _step(meta, 150, "fp32"), _step(meta, 150, mode)
    rel = lambda x, y: ((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30)).item()
    stats = dict(loss=abs(b["loss"] - a["loss"]) / abs(a["loss"]), feat=rel(b["feat"], a["feat"]), mt=0.0, running=0.0)
    for k in a["mt"]:
        stats["mt"] = max(stats["mt"], float((b["mt"][k] - a["mt"][k]).abs().max() / a["mt"][k].abs().max().clamp_min(1e-6)))
    for k, v in a["state"].items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            stats["running"] = max(stats["running"], rel(b["state"][k], v))
    cos_by_block = {}
    for k, g in a["grads"].items():
        if g.numel() < 256 or float(g.norm()) == 0.0:
            continue
        cos = float((g.double().flatten() @ b["grads"][k].double().flatten()) / (g.double().norm() * b["grads"][k].double().norm()).clamp_min(1e-30))
        blk = k.split(".")[1] if k.startswith("convnet.") else "heads"
        cos_by_block[blk] = min(cos_by_block.get(blk, 1.0), cos)
    gn = lambda d: float(torch.sqrt(sum((v.double() ** 2).sum() for v in d.values())))
    stats["grad_norm"] = abs(gn(b["grads"]) - gn(a["grads"])) / gn(a["grads"])
    print(f"{mode} vs fp32 (cfg={cfg}, B={B}): " + ", ".join(f"{k} {v:.2e}" for k, v in stats.items()))
    print("   min gradient cosine per block: " + ", ".join(f"{k} {v:.3f}" for k, v in cos_by_block.items()))
    assert stats["loss"] < 5e-3 and stats["feat"] < 8e-2 and stats["mt"] < 0.25 and stats["running"] < 2e-2, stats
    assert stats["grad_norm"] < 0.2, stats
    assert cos_by_block["heads"] > 0.99 and min(cos_by_block.values()) > 0.5, cos_by_block
