"""Is the backbone forward (and the data-gradient chain) bitwise reproducible run to run?"""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import refmodel as R
from oracle.synth import make_inputs, make_state
from trackertraincode.backbones.mobilenet_v1 import MobileNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = {k: v for k, v in R.state_shapes(False, False).items() if k.startswith("convnet.")}
sd = make_state(shapes, 0)
image, _ = make_inputs(B, seed=7)
G = torch.from_numpy(np.random.default_rng(5).standard_normal((B, 1024)).astype(np.float32)).cuda()
outs = []
for rep in range(3):
    net = MobileNet(num_classes=None).cuda()
    net.load_state_dict({k[len("convnet."):]: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    net.train()
    x = torch.from_numpy(image).cuda().requires_grad_(False)
    f = net.forward_features(x)
    (f * G).sum().backward()
    torch.cuda.synchronize()
    outs.append((f.detach().cpu(), {k: p.grad.cpu() for k, p in net.named_parameters()}))
for rep in (1, 2):
    print("features bitwise equal:", torch.equal(outs[0][0], outs[rep][0]))
    diff = {k: float((outs[0][1][k] - outs[rep][1][k]).abs().max() / outs[0][1][k].abs().max().clamp_min(1e-30)) for k in outs[0][1]}
    worst = sorted(diff.items(), key=lambda kv: -kv[1])[:4]
    print("  grads max rel diff (worst):", [(k, f"{v:.1e}") for k, v in worst])
