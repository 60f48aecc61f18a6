import os, sys, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from util import build_net, load_golden, make_batches, script_args, train_script
import trackertraincode.train as train
cfg = os.environ.get("CFG", "full")
d, meta = load_golden(f"model_{cfg}.npz")
S = train_script()
net = build_net(meta, "cuda").train()
crit, _ = S.setup_losses(script_args(meta["flags"]), net)
opt, sch = S.create_optimizer(net, script_args(meta["flags"], epochs=20))
batches = make_batches(meta, "cuda")
mode = os.environ.get("MODE", "all")
def eager():
    opt.zero_grad(set_to_none=True)
    out = train.training_step(net, batches, 0, crit); out["loss"].backward(); opt.step(); return out
eager(); torch.cuda.synchronize(); print("eager ok", flush=True)
opt.sync_hyper_to_device()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    if mode == "fwd":
        out = train.training_step(net, batches, 0, crit)
    elif mode == "fwdbwd":
        out = train.training_step(net, batches, 0, crit); out["loss"].backward()
    else:
        out = train.training_step(net, batches, 0, crit); out["loss"].backward(); opt.step()
print("captured", mode, flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed", float(out["loss"]), flush=True)
