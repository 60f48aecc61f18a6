"""Run one pointwise-conv entry point repeatedly on one shape (for rocprofv3 --pmc). usage: one_gemm.py kind M K N reps"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
kind, M, ci, co, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = "cuda"
ydw, y, g = torch.randn(M, ci, device=dev), torch.randn(M, co, device=dev), torch.randn(M, co, device=dev)
w, wt = torch.randn(co, ci, device=dev) * 0.05, torch.randn(ci, co, device=dev) * 0.05
bn_dw, bn_pw = torch.rand(8, ci, device=dev) + 0.5, torch.rand(8, co, device=dev) + 0.5
out, gdw, dW = torch.empty(M, co, device=dev), torch.empty(M, ci, device=dev), torch.zeros(co, ci, device=dev)
wq = torch.empty(3 * 1024 * 1024, dtype=torch.int16, device=dev)
part = torch.empty(L.partial_rows_gemm(M) * 2 * max(ci, co), device=dev)
for _ in range(reps):
    if kind == "fwd": L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), p(w), p(out), p(part), M, ci, co, p(wq))
    elif kind == "dgrad": L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bn_pw), p(wt), p(ydw), p(bn_dw), p(gdw), p(part), M, ci, co, p(wq))
    else: L.call("ttk_pwconv1x1_bwd_weight", p(g), p(y), p(bn_pw), p(ydw), p(bn_dw), p(dW), M, ci, co)
torch.cuda.synchronize()
