"""Diagnostic (GPU box): per-parameter gradient error of HIP vs oracle-fp32, both against oracle-fp64."""
import sys, os
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
from oracle import refmodel as R
from oracle.synth import make_inputs, make_state
from trackertraincode.backbones.mobilenet_v1 import MobileNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = {k: v for k, v in R.state_shapes(False, False).items() if k.startswith("convnet.")}
sd = make_state(shapes, 0)
image, _ = make_inputs(B, seed=7)
G = np.random.default_rng(5).standard_normal((B, 1024)).astype(np.float32)

def run_oracle(dtype):
    st = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.array(v))
        if t.is_floating_point(): t = t.to(dtype)
        if not R.is_buffer(k): t.requires_grad_(True)
        st[k] = t
    feat, _ = R.mobilenet_forward(st, torch.from_numpy(image).to(dtype), True)
    (feat * torch.from_numpy(G).to(dtype)).sum().backward()
    return feat.detach(), {k: v.grad for k, v in st.items() if v.grad is not None}

f64, g64 = run_oracle(torch.float64)
f32, g32 = run_oracle(torch.float32)
net = MobileNet(num_classes=None).cuda()
net.load_state_dict({k[8:]: torch.from_numpy(np.array(v)) for k, v in sd.items()})
net.train()
feat = net.forward_features(torch.from_numpy(image).cuda())
(feat * torch.from_numpy(G).cuda()).sum().backward()
rel = lambda a, b: ((a.double().flatten() - b.double().flatten()).norm() / b.double().norm().clamp_min(1e-30)).item()
print("feat: hip %.2e  cpu32 %.2e" % (rel(feat.detach().cpu(), f64), rel(f32, f64)))
for k, p in net.named_parameters():
    print("%-28s hip %.2e   cpu32 %.2e   |g| %.3e" % (k, rel(p.grad.cpu(), g64["convnet." + k]), rel(g32["convnet." + k], g64["convnet." + k]), g64["convnet."+k].norm().item()))
