"""Stress: the first step of the golden network at batch B, repeated N times from the SAME state - every module output's checksum, every per-sample loss and every
gradient must repeat (deterministic mode: no order-dependent reduction anywhere).  Run several copies side by side to perturb the timing (several processes
time-slice the GPU: kernels get preempted).   TTK_DETERMINISTIC=1 python tools/debug/forward_repeat.py [cfg] [B] [N]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p_ in (REPO, REPO + "/neuralnet-tracker-traincode_amd", REPO + "/tests"):
    sys.path.insert(0, p_)
import torch
from util import build_net, load_golden, make_batches, script_args, train_script
import trackertraincode.train as train
import trackertraincode.backbones.mobilenet_v1 as MB

if os.environ.get("POISON"):  # every torch.empty / empty_like of a floating dtype comes back filled with NaN: a kernel that leaves elements of its output
    _e, _el = torch.empty, torch.empty_like  # unwritten shows up as NaN downstream (single process, deterministic)
    def _pe(*a, **k):
        t = _e(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() and t.is_cuda else t
    def _pel(*a, **k):
        t = _el(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() and t.is_cuda else t
    torch.empty, torch.empty_like = _pe, _pel
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 300
_, meta = load_golden(f"model_{cfg}.npz")
meta = dict(meta, B=B, split=(B * 5) // 8)
S = train_script()
if os.environ.get("PRECISION"):  # PRECISION=bf16-compute
    MB.set_activation_dtype(os.environ["PRECISION"])
net = build_net(meta, "cuda").train()
crit, _ = S.setup_losses(script_args(meta["flags"]), net)
batches = make_batches(meta, "cuda")
state0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
names, sums = [], []
import ctypes
import trackertraincode._hip as H
dump = None
if hasattr(H.lib().cdll, "ttk_debug_set_heads_dump"):  # a -DTTK_HEADS_DBG build (TTK_LIB=...): per-lane intermediates of heads_fwd_k
    dump = torch.zeros(B, 64, 16, device="cuda")
    H.lib().cdll.ttk_debug_set_heads_dump.argtypes = [ctypes.c_void_p]
    assert H.lib().cdll.ttk_debug_set_heads_dump(dump.data_ptr()) == 0
def cs(name, t):
    if torch.is_tensor(t) and t.is_floating_point():
        names.append(name); sums.append(torch.stack([t.double().sum(), t.double().abs().sum()]))
keep = {}
def walk(name, o):
    if torch.is_tensor(o):
        cs(name, o)
        if name == "fwd:net.pt3d_68":
            keep["pts"] = o.detach().clone()
            if dump is not None: keep["dump"] = dump.clone()
    elif isinstance(o, dict):
        for k, v in o.items(): walk(f"{name}.{k}", v)
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o): walk(f"{name}[{i}]", v)
for mname, m in net.named_modules():
    m.register_forward_hook(lambda mod, inp, out, mname=mname: walk("fwd:" + (mname or "net"), out))
MB._EXP_TENSOR_HOOK = lambda kind, block, t: cs(f"{kind}{block}", t)
ref = None
bad = 0
for it in range(N):
    net.load_state_dict(state0)
    names.clear(); sums.clear()
    for p in net.parameters(): p.grad = None
    out = train.training_step(net, batches, 150, crit)
    for k, v in out["mt_losses"].items(): cs("loss:" + k, v)
    out["loss"].backward()
    for k, p in net.named_parameters():
        if p.grad is not None: cs("grad:" + k, p.grad)
    cur = (list(names), torch.stack(sums).cpu())
    if os.environ.get("POISON"):
        nanl = [n for n, v in zip(cur[0], cur[1][:, 0].tolist()) if v != v]
        print(f"iteration {it}: non-finite checksums: {nanl[:20]}", flush=True)
    if ref is None:
        ref = cur
        keep["ref"] = keep["pts"]
        keep["dref"] = keep.get("dump")
        continue
    assert cur[0] == ref[0]
    d = (cur[1][:, 0] - ref[1][:, 0]).abs() / ref[1][:, 1].clamp_min(1e-30)
    idx = (d > 1e-12).nonzero().flatten().tolist()
    if idx:
        bad += 1
        dd = (keep["pts"] != keep["ref"])
        w = dd.nonzero()
        print(f"   pt3d_68: {len(w)} elements differ, samples {sorted(set(w[:, 0].tolist()))[:10]}, points {sorted(set(w[:, 1].tolist()))[:70]}; e.g. {keep['pts'][dd][:6].tolist()} vs {keep['ref'][dd][:6].tolist()}")
        if dump is not None:
            dm = (keep["dump"] != keep["dref"])
            wd = dm.nonzero()
            cols = sorted(set(wd[:, 2].tolist()))
            print(f"   dump: {len(wd)} entries differ; lanes {sorted(set(wd[:, 1].tolist()))}; columns {cols} (0-2 local, 3-6 qk, 7-9 ck, 10-11 sh, 12 kp, 13 eig, 14-15 out)")
            for (ss, ll, cc) in wd[:6].tolist():
                print(f"      sample {ss} lane {ll} col {cc}: {keep['dump'][ss, ll, cc].item():.7g} vs {keep['dref'][ss, ll, cc].item():.7g}")
        other = [cur[0][i] for i in idx if not (cur[0][i].startswith("grad:convnet") or (cur[0][i][0] == "g" and cur[0][i][1:].isdigit()))]
        print(f"   outside the backbone: {other[:12]}")
        print(f"iteration {it}: {len(idx)} checksums differ; first: " + "; ".join(f"{cur[0][i]} ({d[i]:.2e})" for i in idx[:6]), flush=True)
print(f"{cfg} B={B}: {N} iterations, {bad} deviating (deterministic={os.environ.get('TTK_DETERMINISTIC', '0')})")
