"""Soak run: N optimiser steps of the benchmark's training step on a fixed synthetic batch set (it over-fits), printing the
loss every 50 steps - finite losses that fall show that the fp16-split operand bounds hold up as weights and activation
statistics drift over hundreds of steps, not only on fresh networks.   python tools/soak.py [--backbone resnet18] [--steps 600]"""
import argparse
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="mobilenetv1")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--blurpool", action="store_true")
ap.add_argument("--precision", default="fp32", help="fp32 | bf16-compute (mobilenetv1)")
a = ap.parse_args()
sys.argv = ["bench.py", "--backbone", a.backbone, "--batch", str(a.batch), "--precision", a.precision] + (["--blurpool"] if a.blurpool else [])
args = bench.parse()
dev = torch.device("cuda", 0)
import trackertraincode.backbones.mobilenet_v1 as MB  # noqa: E402

MB.set_activation_dtype(a.precision)
net, crit, opt, batches, train = bench.build_step(args, dev)
params = list(net.parameters())
first = last = None
for it in range(a.steps):
    for p in params:
        p.grad = None
    out = train.training_step(net, batches, 0, crit)
    out["loss"].backward()
    opt.step()
    if it % 50 == 0 or it == a.steps - 1:
        v = float(out["loss"])
        gmax = max(float(p.grad.abs().max()) for p in params if p.grad is not None)
        print(f"step {it:4d}  loss {v:.5f}  max|grad| {gmax:.3e}  allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB", flush=True)
        assert math.isfinite(v) and math.isfinite(gmax), "non-finite loss or gradient"
        first = v if first is None else first
        last = v
print("first", first, "last", last)
assert last < 0.7 * first, "the loss did not fall"
print("soak ok")
