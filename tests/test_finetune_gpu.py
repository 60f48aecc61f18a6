"""Fine-tuning with frozen BatchNorm statistics (reference models.py:378-394 prepare_finetune / train, modelcomponents.py:208-215
freeze_norm_stats): the normalisation layers stay in eval mode with frozen affine parameters while the convolutions train.
The HIP backbones run their backward through the fixed affine maps (ttk_bn_bwd_frozen); checked against the CPU oracle
evaluated with eval-mode BatchNorm under torch autograd (fp64 = exact arithmetic, fp32 = the reference's own accuracy)."""
import numpy as np
import pytest
import torch

from oracle import refmodel as R
from oracle.synth import make_inputs, make_state

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _state(shapes, seed):
    sd = make_state(shapes, seed)
    rng = np.random.default_rng(seed + 1)
    for k in sd:  # non-trivial running statistics (a trained network's): means of a few tenths, variances around one
        if k.endswith("running_mean"):
            sd[k] = (rng.standard_normal(sd[k].shape) * 0.2).astype(np.float32)
        elif k.endswith("running_var"):
            sd[k] = (0.5 + rng.random(sd[k].shape)).astype(np.float32)
        elif k.endswith("bn2.weight"):  # ResNet18's zero-initialised residual scales would hide the second convolutions
            sd[k] = (0.5 + rng.random(sd[k].shape)).astype(np.float32)
    return sd


def _oracle(forward, sd, image, G, dtype, prefix):
    st = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.array(v))
        t = t.to(dtype) if t.is_floating_point() else t
        if not R.is_buffer(k) and t.dim() == 4:  # the convolutions train; BatchNorm weight / bias are frozen
            t.requires_grad_(True)
        st[k] = t
    feat = forward(st, torch.from_numpy(image).to(dtype), False)[0]
    (feat * torch.from_numpy(G).to(dtype)).sum().backward()
    return feat.detach(), st


def _check(net, sd, forward, F, B, prefix):
    from trackertraincode.neuralnets.modelcomponents import freeze_norm_stats
    image, _ = make_inputs(B, seed=11)
    G = np.random.default_rng(3).standard_normal((B, F)).astype(np.float32)
    f64, st64 = _oracle(forward, sd, image, G, torch.float64, prefix)
    f32, st32 = _oracle(forward, sd, image, G, torch.float32, prefix)
    net.load_state_dict({k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    net.train()
    net.apply(freeze_norm_stats)
    feat = net.forward_features(torch.from_numpy(image).cuda())
    (feat * torch.from_numpy(G).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(feat.detach().cpu(), f64) < 3 * _rel(f32, f64) + 2e-5
    for k, v in net.state_dict().items():  # frozen statistics: nothing moves
        assert torch.equal(v, before[k]), k
    n = 0
    for k, p_ in net.named_parameters():
        if p_.dim() != 4:
            assert p_.grad is None, k  # BatchNorm weight / bias: frozen
            continue
        g64 = st64[prefix + k].grad
        e_hip, e_cpu = _rel(p_.grad.cpu(), g64), _rel(st32[prefix + k].grad, g64)
        # no batch statistics -> no chaotic amplification: a fixed bound next to the relative criterion of the training tests
        assert e_hip < 3 * e_cpu + 3e-5, (k, e_hip, e_cpu)
        n += 1
    return n


@pytest.mark.parametrize("B,blur", [(3, False), (8, False), (8, True)])
def test_mobilenet_frozen_batchnorm_backward(B, blur):
    from trackertraincode.backbones.mobilenet_v1 import MobileNet
    shapes = {k: v for k, v in R.state_shapes(False, False, use_blurpool=blur).items() if k.startswith("convnet.")}
    net = MobileNet(num_classes=None, use_blurpool=blur).cuda()
    assert _check(net, _state(shapes, 2), R.mobilenet_forward, 1024, B, "convnet.") == 27


def test_resnet18_frozen_batchnorm_backward():
    from trackertraincode.backbones.resnet import resnet18
    net = resnet18().cuda()
    assert _check(net, _state(R.resnet18_state_shapes(), 4), R.resnet18_forward, 512, 4, "") == 20


def test_network_finetune_step():
    """prepare_finetune() + train() + one optimiser step with per-group learning rates: BatchNorm layers untouched, every
    convolution and head parameter moves (reference models.py:378-394; ClipAdam takes the 66 groups)."""
    import trackertraincode.train as train
    from trackertraincode.neuralnets.models import NetworkWithPointHead
    torch.manual_seed(0)
    net = NetworkWithPointHead(enable_point_head=False, enable_uncertainty=False).cuda()
    groups = net.prepare_finetune()
    assert len(groups) == 13 * 5 + 1 and sum(len(g) for g in groups) == len(list(net.parameters()))  # per block: conv_dw, bn_dw, conv_sep, bn_sep, relu; the rest
    net.train()
    bns = [m for m in net.convnet.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(bns) == 27 and not any(m.training for m in bns) and net.training
    opt = train.ClipAdam([{"params": [p for p in g if p.requires_grad], "lr": 1e-3 * 0.9 ** i} for i, g in enumerate(reversed(groups))], lr=1e-3)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.rand(6, 1, 129, 129, device="cuda") - 0.5
    out = net(x)
    loss = out["coord"].square().mean() + out["roi"].square().mean() + out["rot"].value.square().mean()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    for k, v in net.state_dict().items():
        is_bn = ".bn" in k and k.startswith("convnet.")
        if is_bn or "num_batches_tracked" in k:
            assert torch.equal(v, before[k]), k
        elif k.startswith("convnet.") and k.endswith("weight"):
            assert not torch.equal(v, before[k]), k
