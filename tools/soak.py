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
ap.add_argument("--seed", type=int, default=0, help="weight initialisation (run-to-run spread of the legs)")
ap.add_argument("--round", default="", help="fp32 experiment: comma list of g:LO-HI / y:LO-HI / yc:LO-HI (centred) - round the gradients / raw conv outputs of blocks LO..HI (stem "
                "side = 0, 13 = the pooled gradient) to the bf16 grid in place after their kernels (profiles/r06_soak_rounding_ab.txt)")
a = ap.parse_args()
sys.argv = ["bench.py", "--backbone", a.backbone, "--batch", str(a.batch), "--precision", a.precision] + (["--blurpool"] if a.blurpool else [])
args = bench.parse()
args.seed = a.seed
dev = torch.device("cuda", 0)
import trackertraincode.backbones.mobilenet_v1 as MB  # noqa: E402

MB.set_activation_dtype(a.precision)
if a.round:
    assert a.precision == "fp32", "--round instruments the fp32 path"
    spans = {}
    for item in a.round.split(","):
        kind, span = item.split(":")
        lo, hi = (int(v) for v in span.split("-"))
        spans[kind] = (lo, hi)

    def _round(kind, block, t):
        lo, hi = spans.get(kind, (1, 0))
        if lo <= block <= hi:
            t.copy_(t.to(torch.bfloat16))
        lo, hi = spans.get(kind + "c", (1, 0))  # "yc" / "gc": CENTRED rounding - the per-channel batch mean is taken out before the rounding and put back after it
        if lo <= block <= hi:
            t3 = t.view(t.shape[-1] // 32, -1, 32)  # the channel-block layout [C/32][pixels][32] under the logical shape
            m = t3.mean(1, keepdim=True)
            t3.copy_((t3 - m).to(torch.bfloat16).float() + m)
    MB._EXP_TENSOR_HOOK = _round
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
