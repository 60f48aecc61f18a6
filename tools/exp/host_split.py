"""Where the eager step's host time goes (wall clock of the enqueueing thread, no synchronisation inside): python tools/exp/host_split.py [--batch 512]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import torch
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
net, crit, opt, batches, train = bench.build_step(args, dev)
import trackertraincode.backbones.mobilenet_v1 as MB
if os.environ.get("CHECKED_PTR"):  # A/B: every pointer through the checked _hip.ptr again (the path until round 6)
    MB._hip.fast_ptr = MB._hip.ptr
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
wrap(MB, "_forward_impl", "backbone_fwd"); wrap(MB, "_backward_impl", "backbone_bwd")
params = list(net.parameters())
def step():
    t0 = time.perf_counter()
    for p in params: p.grad = None
    t1 = time.perf_counter()
    out = train.training_step(net, batches, 0, crit)
    t2 = time.perf_counter()
    out["loss"].backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, t4 - t3
for _ in range(10): step()
torch.cuda.synchronize(); T.clear()
acc = [0.0] * 4; N = 50
for _ in range(N):
    torch.cuda.synchronize()
    for i, v in enumerate(step()): acc[i] += v
torch.cuda.synchronize()
names = ["zero_grad", "training_step (fwd + loss)", "backward", "optimizer"]
for n, v in zip(names, acc): print(f"{n:30s} {v / N * 1e3:7.3f} ms")
for k, v in T.items(): print(f"   of which {k:20s} {v / N * 1e3:7.3f} ms")
print(f"{'total':30s} {sum(acc) / N * 1e3:7.3f} ms")
