"""GPU micro-benchmark of the three pointwise-conv entry points on the layer shapes (and a big square one)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shapes = [("dw2_1", 65, 32, 64), ("dw2_2", 33, 64, 128), ("dw3_1", 33, 128, 128), ("dw3_2", 17, 128, 256), ("dw4_1", 17, 256, 256),
          ("dw4_2", 9, 256, 512), ("dw5_x", 9, 512, 512), ("dw5_6", 5, 512, 1024), ("dw6", 5, 1024, 1024), ("big", 0, 1024, 1024)]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for name, hw, ci, co in shapes:
    M = B * hw * hw if hw else 65536
    dev = "cuda"
    ydw, y, g = torch.randn(M, ci, device=dev), torch.randn(M, co, device=dev), torch.randn(M, co, device=dev)
    w, wt = torch.randn(co, ci, device=dev) * 0.05, torch.randn(ci, co, device=dev) * 0.05
    bn_dw, bn_pw = torch.rand(8, ci, device=dev) + 0.5, torch.rand(8, co, device=dev) + 0.5
    out, gdw, dW = torch.empty(M, co, device=dev), torch.empty(M, ci, device=dev), torch.zeros(co, ci, device=dev)
    wq = torch.empty(3 * 1024 * 1024, dtype=torch.int16, device=dev)
    part = torch.empty(L.partial_rows_gemm(M) * 2 * max(ci, co), device=dev)
    calls = {
        "fwd": lambda: L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), p(w), p(out), p(part), M, ci, co, p(wq)),
        "dgrad": lambda: L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bn_pw), p(wt), p(ydw), p(bn_dw), p(gdw), p(part), M, ci, co, p(wq)),
        "wgrad": lambda: L.call("ttk_pwconv1x1_bwd_weight", p(g), p(y), p(bn_pw), p(ydw), p(bn_dw), p(dW), M, ci, co),
    }
    line = f"{name:6s} M={M:8d} K={ci:4d} N={co:4d} "
    for k, fn in calls.items():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tf = 2 * M * ci * co / us / 1e6
        line += f"| {k} {us:7.1f} us {tf:6.1f} TF "
        if name != "big": tot[k] += us * (5 if name == "dw5_x" else 1)
    print(line)
print("per-step totals (us):", {k: round(v) for k, v in tot.items()}, "ideal each:", round(sum(2*B*hw*hw*ci*co*(5 if n=='dw5_x' else 1) for n,hw,ci,co in shapes if hw)/157.3e6))
