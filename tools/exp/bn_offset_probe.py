"""How far are the raw convolution outputs from zero mean, in units of their standard deviation?  (bf16 storage rounds y at 2^-9 |y|: with
|mean| = r sigma the noise in the NORMALISED activation is ~sqrt(1 + r^2) x what a centred tensor would carry.)  Prints |running_mean| /
sqrt(running_var) per BatchNorm layer of the MobileNet backbone after N steps of the soak loop.   python tools/exp/bn_offset_probe.py [steps] [precision]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
sys.argv = ["bench.py", "--precision", prec]
args = bench.parse()
dev = torch.device("cuda", 0)
import trackertraincode.backbones.mobilenet_v1 as MB  # noqa: E402

MB.set_activation_dtype(prec)
net, crit, opt, batches, train = bench.build_step(args, dev)
params = list(net.parameters())
for it in range(steps):
    for p in params:
        p.grad = None
    out = train.training_step(net, batches, 0, crit)
    out["loss"].backward()
    opt.step()
sd = net.state_dict()
tot_n = tot_w = 0.0
for k in sd:
    if k.endswith("running_mean") and k.startswith("convnet"):
        m, v = sd[k].double(), sd[k.replace("running_mean", "running_var")].double()
        r = (m.abs() / v.sqrt().clamp_min(1e-12))
        amp = (1 + r * r).sqrt()
        print(f"{k:60s} C={m.numel():5d}  |mean|/sigma median {r.median().item():.2f}  mean {r.mean().item():.2f}  max {r.max().item():.2f}   noise amplification sqrt(1+r^2): mean {amp.mean().item():.2f}")
        tot_n += m.numel()
        tot_w += amp.sum().item()
print(f"after {steps} steps ({prec}), loss {float(out['loss']):.4f}: channel-mean amplification {tot_w / tot_n:.2f}")
