"""GPU parity of the HIP ResNet18 backbone (through the C-ABI) against the CPU oracle.  The oracle restates
torchvision's BasicBlock (parity unpinned: torchvision is absent here and not vendored by the reference)."""
import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.synth import make_inputs, make_state

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _run_oracle(sd, image, G, dtype):
    st = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.array(v))
        if t.is_floating_point():
            t = t.to(dtype)
        if not R.is_buffer(k):
            t.requires_grad_(True)
        st[k] = t
    feat, _ = R.resnet18_forward(st, torch.from_numpy(image).to(dtype), True)
    (feat * torch.from_numpy(G).to(dtype)).sum().backward()
    return feat.detach(), st


def _one_fwd_bwd(B, seed, sd, blur=False):
    """-> (strictly ok?, worst relative gradient error); asserts the forward criteria."""
    from trackertraincode.backbones.resnet import resnet18

    image, _ = make_inputs(B, seed=seed)
    G = np.random.default_rng(5).standard_normal((B, 512)).astype(np.float32)
    f64, st64 = _run_oracle(sd, image, G, torch.float64)
    f32, st32 = _run_oracle(sd, image, G, torch.float32)
    net = resnet18(use_blurpool=blur).cuda()
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    net.train()
    feat, none = net(torch.from_numpy(image).cuda())
    assert none is None and feat.shape == (B, 512)
    (feat * torch.from_numpy(G).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(feat.detach().cpu(), f64) < 3 * _rel(f32, f64) + 2e-5
    assert _rel(feat.detach().cpu(), f32) < 1e-4
    for k, v in net.state_dict().items():
        ref = st32[k].detach()
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref) == 1
        elif "running_" in k:
            np.testing.assert_allclose(v.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=1e-6, err_msg=k)
    bad, errs = [], []
    for k, p_ in net.named_parameters():
        g64 = st64[k].grad
        assert p_.grad is not None, k
        e_hip, e_cpu = _rel(p_.grad.cpu(), g64), _rel(st32[k].grad, g64)
        errs.append(e_hip)
        if e_hip > max(3 * e_cpu, 1e-4):
            bad.append((k, e_hip, e_cpu))
    return not bad, max(errs), float(np.median(errs)), bad


@pytest.mark.parametrize("B,blur", [(3, False), (8, False), (3, True), (8, True)])
def test_resnet18_train_fwd_bwd_matches_oracle_parity_unpinned(B, blur):
    """Small batches, ONE input (no retry over inputs).  Forward: as close to the fp64 oracle as the fp32 CPU oracle is (factor 3 +
    2e-5).  Gradients: held to max(3 x the fp32 CPU oracle's distance to fp64, 1e-4) per parameter - with explicit accounting for what a
    small batch cannot avoid: a ReLU / max-pool decision whose pre-activation lies within rounding of zero legitimately differs between
    two fp32 evaluations, and ONE flipped element of a [B,5,5,512] tensor shifts the gradient of every layer upstream by
    ~1/sqrt(elements) ~ 5e-3 relative at B = 8.  A parameter that misses the strict bound is a "flip" and must (a) stay inside the
    flip-sized bound 3e-2, and (b) belong to a run in which the TYPICAL (median) parameter still meets 5e-3; the flips are printed.
    The arithmetic itself is held to 1.5e-6 per kernel in test_conv_gpu.py; B = 512 (below) has no flip allowance beyond single BatchNorm
    channels.  "parity unpinned": the oracle restates torchvision's BasicBlock (module docstring)."""
    # blur: use_blurpool=True - BlurPool2D in front of every block's first convolution and in the max-pool's place (resnet.py:31-49,63-66)
    sd = make_state(R.resnet18_state_shapes(use_blurpool=blur), seed=0)
    ok, worst, median, bad = _one_fwd_bwd(B, 7, sd, blur)
    print(f"resnet18 B={B} blur={blur}: worst gradient rel {worst:.1e}, median {median:.1e}, parameters beyond the strict bound: {[(k, f'{a:.1e}', f'{b:.1e}') for k, a, b in bad]}")
    assert worst < 3e-2, (worst, bad[:4])
    assert ok or median < 5e-3, (median, bad[:4])


def test_resnet18_at_benchmark_size_matches_oracle():
    """BASELINE config 3's size (B = 512), ONE input, the criterion the MobileNet step is held to at this size
    (tests/test_fullsize_gpu.py): every parameter gradient as close to the fp64 oracle as the fp32 CPU oracle is
    (3 x its error + 1e-5) - no retry over inputs, no trimming of a share of the elements; a BatchNorm weight / bias gradient
    may exceed the bound through at most two of its channels (single ReLU decisions, see below), which are printed.  At this batch a ReLU / max-pool decision that
    differs between two fp32 evaluations moves a gradient by 1/64 of what it does at B = 8.  Reference:
    backbones/resnet.py:52-104 (torchvision BasicBlock: parity unpinned, see the module docstring)."""
    import gc
    import os

    from trackertraincode.backbones.resnet import resnet18

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    B = 512
    sd = make_state(R.resnet18_state_shapes(), seed=0)
    image, _ = make_inputs(B, seed=7)
    G = np.random.default_rng(5).standard_normal((B, 512)).astype(np.float32)
    net = resnet18().cuda()
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    net.train()
    feat, none = net(torch.from_numpy(image).cuda())
    assert none is None and feat.shape == (B, 512)
    (feat * torch.from_numpy(G).cuda()).sum().backward()
    torch.cuda.synchronize()
    hip_feat = feat.detach().cpu()
    hip_grads = {k: p_.grad.detach().cpu() for k, p_ in net.named_parameters()}
    hip_state = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    del net, feat
    torch.cuda.empty_cache()

    def oracle(dtype):
        f, st = _run_oracle(sd, image, G, dtype)
        out = f, {k: v.grad for k, v in st.items() if not R.is_buffer(k)}, {k: v.detach() for k, v in st.items() if "running_" in k}
        del st
        gc.collect()
        return out

    f32, g32, run32 = oracle(torch.float32)
    f64, g64, _ = oracle(torch.float64)
    e_feat = _rel(hip_feat, f64)
    assert e_feat < 3 * _rel(f32, f64) + 2e-5, e_feat
    assert _rel(hip_feat, f32) < 1e-4
    for k, ref in run32.items():
        np.testing.assert_allclose(hip_state[k].numpy(), ref.numpy(), rtol=2e-4, atol=1e-6, err_msg=k)
    bad, worst, flips = [], (0.0, ""), []
    for k, g in hip_grads.items():
        e_hip, e_cpu = _rel(g, g64[k]), _rel(g32[k], g64[k])
        if e_hip > worst[0]:
            worst = (e_hip, k)
        if e_hip > 3 * e_cpu + 1e-5:
            # ONE ReLU decision of the last block that falls on the other side of zero (its pre-activation within fp32 rounding of
            # zero: ~6 of the 6.5 M elements are that close) changes ONE channel of that block's BatchNorm weight / bias gradient,
            # a sum of 12 800 terms that cancels to ~1 % of their magnitude - visibly (2e-4 of the tensor), in either fp32
            # implementation.  Signature: the excess sits in at most two channels; without them the tensor meets the bound.
            dev = (g.double() - g64[k].double()).reshape(g.shape[0], -1).pow(2).sum(1)
            top = torch.topk(dev, min(2, dev.numel())).indices
            rest = dev.clone()
            rest[top] = 0.0
            e_rest = (rest.sum().sqrt() / g64[k].double().norm()).item()
            if g.dim() == 1 and e_rest <= 3 * e_cpu + 1e-5:
                flips.append((k, [int(i) for i in top], f"hip {e_hip:.2e} -> {e_rest:.2e} without them", f"cpu32 {e_cpu:.2e}"))
            else:
                bad.append((k, f"hip {e_hip:.2e}", f"cpu32 {e_cpu:.2e}"))
    if flips:
        print("single-decision channels:", flips)
    assert len(flips) <= 4, flips
    print(f"resnet18 B={B}: features rel {e_feat:.1e}, worst gradient rel {worst[0]:.1e} ({worst[1]})")
    assert not bad, bad[:8]


@pytest.mark.parametrize("cfg", ["default", "full"])
def test_resnet18_pose_network_whole_step_at_benchmark_size_parity_unpinned(cfg):
    """BASELINE config 3 as a WHOLE step: NetworkWithPointHead(config="resnet18") at B = 512 through the backbone, the heads, the multi-task
    losses, backward, the global-norm clip and Adam, against the oracle (resnet18_forward + heads_forward + compute_loss + ClipAdam) on
    identical weights and inputs, with the criterion of tests/test_fullsize_gpu.py: loss_sum and every per-sample loss within 1e-3,
    features 1e-4 relative, BatchNorm running statistics 2e-4, every parameter gradient as close to the fp64 oracle as the fp32 CPU path
    is (3 x its error + 1e-5; a BatchNorm vector may exceed it through at most two single-ReLU-decision channels, printed), the global
    gradient norm 1e-3 relative and the parameters after the fused clip + Adam step within 2 lr of the oracle's.  "parity unpinned":
    torchvision is absent from the image and not vendored by the reference - the oracle restates BasicBlock (backbones/resnet.py:52-104,
    neuralnets/models.py:218-232)."""
    import gc
    import itertools
    import os

    import trackertraincode.train as train
    from trackertraincode.neuralnets.models import NetworkWithPointHead
    from test_oracle_golden import _batches, _criterions
    from util import GOLDEN, load_golden, make_batches, script_args, train_script

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    B, epoch = 512, 150
    _, meta = load_golden(f"model_{cfg}.npz")
    meta = dict(meta, B=B, split=(B * 5) // 8)
    meta["config"] = dict(meta["config"], config="resnet18")
    S = train_script()
    net = NetworkWithPointHead(**meta["config"])
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = make_state(shapes, meta["state_seed"])
    # zero_init_residual (resnet.py:101) would make half of every block's gradients trivially zero: unit BatchNorm weights instead
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    net = net.cuda().train()
    crit, _ = S.setup_losses(script_args(meta["flags"]), net)
    opt, _ = S.create_optimizer(net, script_args(meta["flags"], epochs=20))
    lr = opt.param_groups[0]["lr"]
    batches = make_batches(meta, "cuda")
    out = train.training_step(net, batches, epoch, crit)
    out["loss"].backward()
    torch.cuda.synchronize()
    hip_loss = out["loss"].item()
    hip_vals = {k: v.detach().cpu() for k, v in out["mt_losses"].items()}
    hip_grads = {k: (None if p.grad is None else p.grad.detach().cpu().clone()) for k, p in net.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    hip_norm = float(opt.last_grad_norm.item())
    hip_state = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    del net, out, opt
    torch.cuda.empty_cache()

    image, ids = make_inputs(B, seed=meta["input_seed"])
    ocrit, _ = _criterions(meta, GOLDEN)

    def oracle(dtype, step):
        st = {}
        for k, v in sd.items():
            t = torch.from_numpy(np.array(v))
            t = t.to(dtype) if t.is_floating_point() else t
            st[k] = t.requires_grad_(True) if not R.is_buffer(k) else t
        obatches = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()} for b in _batches(meta)]
        o, feat = R.network_forward(st, torch.from_numpy(image).to(dtype), torch.from_numpy(ids), meta["config"], True)
        loss, by_name = R.compute_loss(o, obatches, epoch, ocrit)
        loss.backward()
        res = dict(loss=float(loss.item()), by_name={k: v[0].detach().clone() for k, v in by_name.items()},
                   grads={k: (None if v.grad is None else v.grad.clone()) for k, v in st.items() if not R.is_buffer(k)},
                   running={k: v.detach().clone() for k, v in st.items() if "running_" in k})
        if step:
            oopt = R.ClipAdam(st, lr=1.0e-3, epochs=20)
            res["gnorm"] = float(oopt.step())
            res["lr"] = oopt.lrs()[0]
            res["after"] = {k: v.detach().clone() for k, v in st.items() if not R.is_buffer(k)}
        del st, o, feat, loss, by_name
        gc.collect()
        return res

    o32 = oracle(torch.float32, True)
    assert abs(lr - o32["lr"]) < 1e-12
    assert abs(hip_loss - o32["loss"]) < 1e-3, (hip_loss, o32["loss"])
    assert list(hip_vals.keys()) == list(o32["by_name"].keys())
    for n, v in o32["by_name"].items():
        np.testing.assert_allclose(hip_vals[n].numpy(), v.numpy(), rtol=1e-3, atol=1e-3, err_msg=n)
    for k, v in o32["running"].items():
        np.testing.assert_allclose(hip_state[k].numpy(), v.numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    assert abs(hip_norm - o32["gnorm"]) < 1e-3 * o32["gnorm"], (hip_norm, o32["gnorm"])
    worst_move = max(float((hip_state[k].double() - v.double()).abs().max()) for k, v in o32["after"].items())
    assert worst_move <= 2.02 * lr, (worst_move, lr)
    o64 = oracle(torch.float64, False)
    bad, flips, worst = [], [], (0.0, "")
    for k, g in hip_grads.items():
        g64 = o64["grads"][k]
        if g64 is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        e_hip, e_cpu = _rel(g, g64), _rel(o32["grads"][k], g64)
        if e_hip > worst[0]:
            worst = (e_hip, k)
        if e_hip > 3 * e_cpu + 1e-5:
            dev = (g.double() - g64.double()).reshape(g.shape[0], -1).pow(2).sum(1)
            top = torch.topk(dev, min(2, dev.numel())).indices
            rest = dev.clone()
            rest[top] = 0.0
            e_rest = (rest.sum().sqrt() / g64.double().norm()).item()
            if g.dim() == 1 and e_rest <= 3 * e_cpu + 1e-5:
                flips.append((k, [int(i) for i in top], f"hip {e_hip:.2e} -> {e_rest:.2e} without them", f"cpu32 {e_cpu:.2e}"))
            else:
                bad.append((k, f"hip {e_hip:.2e}", f"cpu32 {e_cpu:.2e}"))
    if flips:
        print("single-decision channels:", flips)
    print(f"resnet18 pose network cfg={cfg} B={B}: loss {hip_loss:.6f} (oracle {o32['loss']:.6f}), grad norm {hip_norm:.4f} ({o32['gnorm']:.4f}), "
          f"worst gradient rel {worst[0]:.1e} ({worst[1]}), largest parameter distance after the step {worst_move / lr:.2f} lr")
    assert len(flips) <= 4, flips
    assert not bad, bad[:8]


def test_resnet18_pose_network_step():
    """NetworkWithPointHead(config='resnet18') trains through the HIP path (reference models.py:221-222, 405)."""
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    torch.manual_seed(0)
    net = NetworkWithPointHead(enable_point_head=False, config="resnet18", backbone_args={"use_blurpool": False}).cuda().train()
    x = torch.rand(4, 1, 129, 129, device="cuda") - 0.5
    out = net(x, torch.zeros(4, dtype=torch.int32, device="cuda"))
    assert out["coord"].shape == (4, 3) and out["roi"].shape == (4, 4)
    (out["coord"].sum() + out["roi"].sum() + out["rot"].value.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.convnet.parameters())


def test_resnet18_eval_forward_matches_oracle():
    """Eval mode (running statistics) on the HIP kernels against the oracle with calibrated statistics."""
    from trackertraincode.backbones.resnet import resnet18

    sd = make_state(R.resnet18_state_shapes(), seed=0)
    image, _ = make_inputs(4, seed=9)
    st = R.state_from_numpy(sd, requires_grad=False)
    with torch.no_grad():
        R.resnet18_forward(st, torch.from_numpy(image), True, momentum=1.0)  # calibrate running stats on this batch
        ref, _ = R.resnet18_forward(st, torch.from_numpy(image), False)
    net = resnet18().cuda()
    net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=True)
    net.eval()
    with torch.no_grad():
        feat, _ = net(torch.from_numpy(image).cuda())
    assert _rel(feat.cpu(), ref) < 2e-4
    with pytest.raises(NotImplementedError):
        net(torch.from_numpy(image).cuda())  # grad mode with trainable parameters in eval: not built


@pytest.mark.parametrize("B,H,W,C,stride", [(3, 33, 33, 64, 1), (2, 65, 65, 64, 2), (5, 17, 17, 128, 2), (4, 9, 9, 256, 1), (2, 6, 5, 32, 2)])
def test_blur3x3_kernels_match_torch(B, H, W, C, stride):
    """ttk_blur3x3_fwd / _bwd on channels-last rows against conv2d with the binomial kernel and its autograd (fp64)."""
    import trackertraincode._hip as H_

    L, p = H_.lib(), H_.ptr
    g = torch.Generator().manual_seed(B * 100 + C)
    a = torch.randn(B, C, H, W, generator=g)
    k = (torch.tensor([1.0, 2.0, 1.0])[:, None] * torch.tensor([1.0, 2.0, 1.0])[None, :] / 16.0).double()
    a64 = a.double().requires_grad_(True)
    t64 = torch.nn.functional.conv2d(a64, k.repeat(C, 1, 1, 1), None, stride=stride, padding=1, groups=C)
    ga, gb = torch.randn(t64.shape, generator=g), torch.randn(t64.shape, generator=g)
    t64.backward((ga + gb).double())
    Ho, Wo = t64.shape[-2:]
    rows = lambda x: x.permute(0, 2, 3, 1).contiguous().cuda()
    a_r = rows(a)
    t = torch.empty((B, Ho, Wo, C), device="cuda")
    L.call("ttk_blur3x3_fwd", p(a_r), p(t), B, H, W, C, stride)
    assert _rel(t.permute(0, 3, 1, 2).cpu(), t64.detach()) < 2e-7
    gin = torch.empty((B, H, W, C), device="cuda")
    ga_r, gb_r = rows(ga), rows(gb)  # (named: a temporary would be recycled before the launch reads it)
    L.call("ttk_blur3x3_bwd", p(ga_r), p(gb_r), p(gin), B, H, W, C, stride)
    assert _rel(gin.permute(0, 3, 1, 2).cpu(), a64.grad) < 3e-7
    L.call("ttk_blur3x3_bwd", p(ga_r), None, p(gin), B, H, W, C, stride)
    a64.grad = None
    torch.nn.functional.conv2d(a64, k.repeat(C, 1, 1, 1), None, stride=stride, padding=1, groups=C).backward(ga.double())
    assert _rel(gin.permute(0, 3, 1, 2).cpu(), a64.grad) < 3e-7


def test_resnet18_blurpool_pose_step_and_eval():
    """NetworkWithPointHead("resnet18", use_blurpool) through heads, losses and clip + Adam, and the eval-mode forward against the oracle."""
    from trackertraincode.backbones.resnet import resnet18
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    net = NetworkWithPointHead(enable_point_head=False, config="resnet18", backbone_args={"use_blurpool": True}).cuda().train()
    assert net.get_config()["backbone_args"] == {"use_blurpool": True}
    x = torch.rand(4, 1, 129, 129, device="cuda") - 0.5
    out = net(x, torch.zeros(4, dtype=torch.int32, device="cuda"))
    (out["coord"].sum() + out["roi"].sum() + out["rot"].value.sum()).backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in net.convnet.parameters())
    sd = make_state(R.resnet18_state_shapes(use_blurpool=True), seed=0)
    image, _ = make_inputs(4, seed=9)
    st = R.state_from_numpy(sd, requires_grad=False)
    with torch.no_grad():
        R.resnet18_forward(st, torch.from_numpy(image), True, momentum=1.0)  # calibrate the running statistics on this batch
        ref, _ = R.resnet18_forward(st, torch.from_numpy(image), False)
    bb = resnet18(use_blurpool=True).cuda()
    bb.load_state_dict({k: v.clone() for k, v in st.items()}, strict=True)
    bb.eval()
    with torch.no_grad():
        feat, _ = bb(torch.from_numpy(image).cuda())
        feat_cpu, _ = bb.cpu()(torch.from_numpy(image))  # the plain-torch module path (export)
    assert _rel(feat.cpu(), ref) < 2e-4 and _rel(feat_cpu, ref) < 1e-5
