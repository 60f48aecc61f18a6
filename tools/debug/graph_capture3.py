import os, sys, faulthandler
faulthandler.enable()
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from util import build_net, load_golden, make_batches, script_args, train_script
import trackertraincode.train as train
d, meta = load_golden("model_full.npz")
S = train_script()
def make():
    net = build_net(meta, "cuda").train()
    crit, _ = S.setup_losses(script_args(meta["flags"]), net)
    opt, sch = S.create_optimizer(net, script_args(meta["flags"], epochs=20))
    return net, crit, opt, sch
batches = make_batches(meta, "cuda")
n_pre = int(os.environ.get("PRE", 6))
if n_pre:
    net_e, crit_e, opt_e, sch_e = make()
    for i in range(n_pre):
        opt_e.zero_grad(set_to_none=True)
        out = train.training_step(net_e, batches, 0, crit_e); out["loss"].backward(); opt_e.step()
        print("eager", i, out["loss"].item(), flush=True)
net_g, crit_g, opt_g, sch_g = make()
g = train.GraphedTrainStep(net_g, crit_g, opt_g)
for i in range(4):
    print("graphed", i, g.run(batches, 0)["loss"].item(), flush=True)
