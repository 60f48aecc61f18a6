"""micro-benchmark of the MobileNet stem forward / weight gradient at B = 512 (tools/exp: scratch)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
B = 512
x = torch.randn(B, 1, 129, 129, device="cuda"); w = torch.randn(32, 1, 5, 5, device="cuda") * 0.1
y = torch.empty(B, 65, 65, 32, device="cuda"); part = torch.empty(L.partial_rows_elementwise(B * 65 * 65 * 8), 2, 32, device="cuda")
def run(fn, name):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, "us", round(e0.elapsed_time(e1) * 100, 1))
run(lambda: L.call("ttk_stem_fwd", p(x), p(w), p(y), p(part), None, B, 129, 129, 0), "stem_fwd")
ref = torch.nn.functional.conv2d(x[:4].double().cpu(), w.double().cpu(), stride=2, padding=2).permute(0, 2, 3, 1)
print("rel err", float((y[:4].cpu().double() - ref).norm() / ref.norm()))
g = torch.randn(B, 65, 65, 32, device="cuda") * 1e-3
bn = torch.rand(8, 32, device="cuda") + 0.5
dw = torch.zeros(32, 25, device="cuda")
run(lambda: L.call("ttk_stem_bwd_weight", p(g), p(y), p(bn), p(x), p(dw), 1, None, B, 129, 129, 0), "stem_wgrad")
