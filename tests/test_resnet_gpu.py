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


@pytest.mark.parametrize("B", [3, 8])
def test_resnet18_train_fwd_bwd_matches_oracle(B):
    """Same criterion as the MobileNet backbone test: as close to the fp64 oracle as the fp32 CPU oracle is
    (factor 3 + 2e-5), robust to single ReLU / max-pool decisions that differ between two fp32 evaluations."""
    from trackertraincode.backbones.resnet import resnet18

    sd = make_state(R.resnet18_state_shapes(), seed=0)
    image, _ = make_inputs(B, seed=7)
    G = np.random.default_rng(5).standard_normal((B, 512)).astype(np.float32)
    f64, st64 = _run_oracle(sd, image, G, torch.float64)
    f32, st32 = _run_oracle(sd, image, G, torch.float32)
    net = resnet18().cuda()
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
    bad = []
    for k, p_ in net.named_parameters():
        g64 = st64[k].grad
        assert p_.grad is not None, k
        e_hip, e_cpu = _rel(p_.grad.cpu(), g64), _rel(st32[k].grad, g64)
        # one ReLU / max-pool decision that differs between two fp32 evaluations (pre-activation within rounding of 0)
        # shifts the gradient of EVERY layer upstream of it by ~1e-4 relative at these tiny batches (B=3 passes at
        # 3*e_cpu; see test_model_gpu.test_gradients_vs_fp64_oracle for the same criterion): 1e-3 is the north-star bound
        if e_hip > max(3 * e_cpu, 1e-3):
            a, b = p_.grad.double().flatten().cpu(), g64.double().flatten()
            dev = (a - b).abs()
            keep = dev <= torch.quantile(dev, 0.99)
            trimmed = (dev[keep].norm() / b.norm().clamp_min(1e-30)).item()
            if trimmed > max(3 * e_cpu, 1e-4):
                bad.append((k, e_hip, e_cpu, trimmed))
    assert not bad, f"gradients further from fp64 than the fp32 CPU path: {bad[:5]}"


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
