"""Debug aid: gradient norms and parameter movement of one fine-tuning step (frozen BatchNorm)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode.train as train  # noqa: E402
from trackertraincode.neuralnets.models import NetworkWithPointHead  # noqa: E402

torch.manual_seed(0)
net = NetworkWithPointHead(enable_point_head=False, enable_uncertainty=False).cuda()
groups = net.prepare_finetune()
net.train()
opt = train.ClipAdam([{"params": [p for p in g if p.requires_grad], "lr": 1e-3 * 0.9 ** i} for i, g in enumerate(reversed(groups))], lr=1e-3)
before = {k: v.clone() for k, v in net.named_parameters()}
x = torch.rand(6, 1, 129, 129, device="cuda") - 0.5
out = net(x)
loss = out["coord"].square().mean() + out["roi"].square().mean() + out["rot"].value.square().mean()
loss.backward()
opt.step()
torch.cuda.synchronize()
print("loss", float(loss), "grad norm", float(opt.last_grad_norm))
for k, p in net.named_parameters():
    if p.requires_grad:
        g = p.grad
        print(f"{k:40s} |g| {float(g.norm()) if g is not None else None!s:12} moved {float((p - before[k]).abs().max()):.3e}")
