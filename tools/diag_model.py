"""Diagnostic (GPU box): whole-network gradient errors vs fp64 oracle, per parameter."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "neuralnet-tracker-traincode_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
from oracle import refmodel as R
from oracle.synth import make_inputs, make_state
from util import GOLDEN, build_net, load_golden, make_batches, script_args, train_script
from test_oracle_golden import _batches
import trackertraincode.train as train
d, meta = load_golden("model_full.npz")
S = train_script()
shapes = {k: tuple(v) for k, v in meta["shapes"].items()}
image, ids = make_inputs(meta["B"], seed=meta["input_seed"])
gmm = R.ShapeGmm(os.path.join(GOLDEN, "shapeparams_gmm.npz"))
fl = meta["flags"]
ocrit, _ = R.setup_losses(with_pointhead=True, with_nll_loss=True, rampup_nll_losses=True, epochs=200, gmm=gmm)
def oracle(dtype):
    st = {}
    for k, v in make_state(shapes, 0).items():
        t = torch.from_numpy(np.array(v)); t = t.to(dtype) if t.is_floating_point() else t
        st[k] = t.requires_grad_(True) if not R.is_buffer(k) else t
    out, feat = R.network_forward(st, torch.from_numpy(image).to(dtype), torch.from_numpy(ids), meta["config"], True)
    feat.retain_grad()
    bs = [{k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()} for b in _batches(meta)]
    loss, _ = R.compute_loss(out, bs, 150, ocrit); loss.backward()
    return {k: v.grad for k, v in st.items() if not R.is_buffer(k)}, feat.grad, feat.detach()
g64, gf64, f64 = oracle(torch.float64); g32, gf32, f32 = oracle(torch.float32)
net = build_net(meta, "cuda").train()
crit, _ = S.setup_losses(script_args(meta["flags"]), net)
hook = {}
orig = net.convnet.forward_features
def ff(x):
    f = orig(x); f.retain_grad(); hook["f"] = f; return f
net.convnet.forward_features = ff
train.training_step(net, make_batches(meta, "cuda"), 150, crit)["loss"].backward()
rel = lambda a, b: ((a.double().flatten().cpu() - b.double().flatten()).norm() / b.double().norm().clamp_min(1e-30)).item()
print("feat  hip %.2e cpu32 %.2e" % (rel(hook["f"].detach(), f64), rel(f32, f64)))
print("gfeat hip %.2e cpu32 %.2e" % (rel(hook["f"].grad, gf64), rel(gf32, gf64)))
for k, p in net.named_parameters():
    if g64[k] is None: continue
    print("%-45s hip %.2e cpu32 %.2e" % (k, rel(p.grad, g64[k]), rel(g32[k], g64[k])))
