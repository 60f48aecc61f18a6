"""Cycle stamps of the weight-gradient producers (libttk_exp20.so, tools/exp/gemm_variants.sh 20)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
H.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libttk_exp20.so"); H._lib = None
L, p = H.lib(), H.ptr
for M, Cin, Cout in ((41472, 512, 512), (12800, 1024, 1024)):
    g = torch.randn(M, Cout, device="cuda"); y = torch.randn(M, Cout, device="cuda"); x = torch.randn(M, Cin, device="cuda")
    bnp = torch.rand(8, Cout, device="cuda") + 0.5; bnd = torch.rand(8, Cin, device="cuda") + 0.5
    dw = torch.zeros(Cout, Cin, device="cuda")
    for _ in range(3):
        L.call("ttk_pwconv1x1_bwd_weight", p(g), p(y), p(bnp), p(x), p(bnd), p(dw), M, Cin, Cout)
    torch.cuda.synchronize()
    names = ["wait A regs", "convert+store A", "issue A loads", "wait B regs", "convert+store B", "issue B loads", "barrier"]
    v = dw.flatten()[:7].cpu().tolist()
    print(M, Cin, Cout, "cycles per k32 step:", ", ".join(f"{n} {c:.0f}" for n, c in zip(names, v)), " total", round(sum(v)))
