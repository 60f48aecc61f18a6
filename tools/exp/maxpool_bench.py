"""micro-benchmark of the ResNet18 stem's 3x3/s2 max-pool forward / backward at B = 512 (tools/exp: scratch)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
B, Hh, C = 512, 65, 64
Ho = (Hh - 1) // 2 + 1
y = torch.randn(B, Hh, Hh, C, device="cuda"); bn = torch.rand(8, C, device="cuda") + 0.5; bn[7] = 0
a = torch.empty(B, Ho, Ho, C, device="cuda"); idx = torch.empty(B, Ho, Ho, C, dtype=torch.uint8, device="cuda")
ga = torch.randn(B, Ho, Ho, C, device="cuda"); g = torch.empty_like(y)
part = torch.empty(L.partial_rows_elementwise(B * Hh * Hh * (C // 4)), 2, C, device="cuda")
def run(fn, name, by):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(name, "us", round(us, 1), "TB/s", round(by / us / 1e6, 2))
run(lambda: L.call("ttk_maxpool3x3s2_fwd", p(y), p(bn), p(a), p(idx), B, Hh, Hh, C), "maxpool_fwd", y.numel() * 4 + a.numel() * 5)
run(lambda: L.call("ttk_maxpool3x3s2_bwd", p(ga), None, p(idx), p(y), p(bn), p(g), p(part), B, Hh, Hh, C), "maxpool_bwd", y.numel() * 8 + a.numel() * 5)
