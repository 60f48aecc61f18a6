import os, sys, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from util import build_net, load_golden, make_batches, script_args, train_script
import trackertraincode.train as train
from trackertraincode.datasets.batch import Batch
d, meta = load_golden("model_full.npz")
S = train_script()
net = build_net(meta, "cuda").train()
crit, _ = S.setup_losses(script_args(meta["flags"]), net)
opt, sch = S.create_optimizer(net, script_args(meta["flags"], epochs=20))
batches = make_batches(meta, "cuda")
V = os.environ.get("V", "a")
def eager():
    opt.zero_grad(set_to_none=True)
    out = train.training_step(net, batches, 0, crit); out["loss"].backward(); opt.step(); return out
if V == "a":      # keep eager output alive across the capture
    keep = eager()
elif V == "b":    # static clones
    eager(); batches = [Batch(b.meta, ((k, v.clone()) for k, v in b.items())) for b in batches]
elif V == "c":    # stream.synchronize instead of device synchronize
    eager(); torch.cuda.current_stream().synchronize()
else:
    eager()
if V != "c": torch.cuda.synchronize()
print("eager ok", flush=True)
opt.sync_hyper_to_device()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    out = train.training_step(net, batches, 0, crit); out["loss"].backward(); opt.step()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed", float(out["loss"]), flush=True)
