"""Fixed cost (prologue + atomic epilogue) of the weight-gradient GEMM: time vs M at 512x512 (32 slices x 8 tiles)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
Cin = Cout = 512
for M in (1024, 2048, 4096, 8192, 16384, 41472):
    g = torch.randn(M, Cout, device="cuda"); y = torch.randn(M, Cout, device="cuda"); x = torch.randn(M, Cin, device="cuda")
    bnp = torch.rand(8, Cout, device="cuda") + 0.5; bnd = torch.rand(8, Cin, device="cuda") + 0.5
    dw = torch.zeros(Cout, Cin, device="cuda")
    f = lambda: L.call("ttk_pwconv1x1_bwd_weight", p(g), p(y), p(bnp), p(x), p(bnd), p(dw), M, Cin, Cout)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(f"M={M:6d}  steps/slice={M/32/32:6.1f}  {s.elapsed_time(e)/20*1e3:7.1f} us")
