import sys, os, torch
sys.path.insert(0, "neuralnet-tracker-traincode_amd")
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
B = 512
x = torch.randn(B, 1, 129, 129, device="cuda"); w = torch.randn(64, 1, 7, 7, device="cuda") * 0.1
y = torch.empty(B, 65, 65, 64, device="cuda"); part = torch.empty(L.partial_rows_elementwise(B * 65 * 65 * 16), 2, 64, device="cuda")
for _ in range(3): L.call("ttk_stem7_fwd", p(x), p(w), p(y), p(part), None, B, 129, 129)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): L.call("ttk_stem7_fwd", p(x), p(w), p(y), p(part), None, B, 129, 129)
e1.record(); torch.cuda.synchronize()
print("stem7_fwd us", e0.elapsed_time(e1) * 100)
ref = torch.nn.functional.conv2d(x[:4].double().cpu(), w.double().cpu(), stride=2, padding=3).permute(0, 2, 3, 1)
print("rel err", float((y[:4].cpu().double() - ref).norm() / ref.norm()))
print("sum check", float((part[:, 0].sum(0).cpu().double() - y.double().sum((0, 1, 2)).cpu()).abs().max()), float(y.double().sum((0,1,2)).abs().max()))
# ---- weight gradient
gr = torch.randn(B, 65, 65, 64, device="cuda") * 1e-2
bn = torch.rand(8, 64, device="cuda") + 0.5
dw = torch.empty(64, 1, 7, 7, device="cuda")
nb = L.cdll.ttk_stem7_wgrad_partial_bytes(B, 129, 129)
scr = torch.empty(nb // 4, device="cuda")
for name, sp in (("atomic", None), ("partials+fold", scr)):
    for _ in range(3): L.call("ttk_stem7_bwd_weight", p(gr), p(y), p(bn), p(x), p(dw), p(sp), B, 129, 129)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): L.call("ttk_stem7_bwd_weight", p(gr), p(y), p(bn), p(x), p(dw), p(sp), B, 129, 129)
    e1.record(); torch.cuda.synchronize()
    print("stem7_bwd_weight", name, "us", round(e0.elapsed_time(e1) * 100, 1))
