"""Quick timing of the HIP backbone fwd+bwd (GPU box)."""
import sys, os, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
from trackertraincode.backbones.mobilenet_v1 import MobileNet
from trackertraincode.backbones.resnet import resnet18
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "mobilenetv1"
torch.manual_seed(0)
net = (resnet18() if which == "resnet18" else MobileNet(num_classes=None)).cuda().train()
if which == "resnet18":
    for m in net.modules():  # zero_init_residual would make half the network's gradients trivially zero
        if hasattr(m, "bn2"): torch.nn.init.constant_(m.bn2.weight, 1.0)
x = torch.rand(B, 1, 129, 129, device="cuda") - 0.5
G = torch.randn(B, net.num_features, device="cuda")
def step():
    for p in net.parameters(): p.grad = None
    f = net.forward_features(x)
    f.backward(G)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.time()
e0, e1, e2 = torch.cuda.Event(True), torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(iters): step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"B={B} fwd+bwd {ms:.2f} ms/step  {B/ms*1000:.0f} crops/s  (wall {(time.time()-t0)/iters*1000:.2f} ms) mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
# fwd only
with torch.no_grad():
    pass
e0.record()
for _ in range(iters):
    f = net.forward_features(x)
e1.record(); torch.cuda.synchronize()
print(f"fwd only {e0.elapsed_time(e1)/iters:.2f} ms")
