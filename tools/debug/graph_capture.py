import os, sys, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
import bench
sys.argv = sys.argv[:1]
args = bench.parse(); args.batch = 64
dev = torch.device("cuda", 0)
net, crit, opt, batches, train = bench.build_step(args, dev)
mode = os.environ.get("MODE", "all")
def eager():
    opt.zero_grad(set_to_none=True)
    out = train.training_step(net, batches, 0, crit); out["loss"].backward(); opt.step(); return out
eager(); torch.cuda.synchronize(); print("eager ok", flush=True)
opt.sync_hyper_to_device()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
x = batches[0]["image"]
with torch.cuda.graph(g):
    if mode == "bbfwd":
        f = net.convnet.forward_features(x)
    elif mode == "bb":
        f = net.convnet.forward_features(x); f.backward(torch.ones_like(f))
    elif mode == "fwd":
        out = train.training_step(net, batches, 0, crit)
    elif mode == "fwdbwd":
        out = train.training_step(net, batches, 0, crit); out["loss"].backward()
    else:
        out = train.training_step(net, batches, 0, crit); out["loss"].backward(); opt.step()
print("captured", mode, flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed", flush=True)
