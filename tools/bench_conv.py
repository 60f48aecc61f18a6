"""GPU micro-benchmark of the three implicit-GEMM convolution entry points on the conv shapes of the ResNet18 variant at
batch B (129x129 input).  python tools/bench_conv.py [B] [iters]     (TTK_GEMM=bf16x3 for the previous kernels)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H  # noqa: E402

L, p = H.lib(), H.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# (name, H, Cin, Cout, k, stride, occurrences per step)
shapes = [("l1", 33, 64, 64, 3, 1, 4), ("l2a", 33, 64, 128, 3, 2, 1), ("l2d", 33, 64, 128, 1, 2, 1), ("l2", 17, 128, 128, 3, 1, 3),
          ("l3a", 17, 128, 256, 3, 2, 1), ("l3d", 17, 128, 256, 1, 2, 1), ("l3", 9, 256, 256, 3, 1, 3),
          ("l4a", 9, 256, 512, 3, 2, 1), ("l4d", 9, 256, 512, 1, 2, 1), ("l4", 5, 512, 512, 3, 1, 3)]
if os.environ.get("EXTRA"):  # the MobileNet dw5_x pointwise shape through the convolution kernels (code-generation comparison)
    shapes = [("pw5", 9, 512, 512, 1, 1, 0)]
tot = {"fwd": 0.0, "dy": 0.0, "dgrad": 0.0, "dgrad_m": 0.0, "wgrad": 0.0}
dev = "cuda"
for name, hw, ci, co, k, s, mult in shapes:
    pad = k // 2
    ho = (hw + 2 * pad - k) // s + 1
    M = B * ho * ho
    a = torch.relu(torch.randn(B, hw, hw, ci, device=dev))
    w = torch.randn(co, ci, k, k, device=dev) * (2.0 / (co * k * k)) ** 0.5
    y, g = torch.randn(B, ho, ho, co, device=dev), torch.randn(B, ho, ho, co, device=dev) * 1e-3
    bn, mbn = torch.rand(8, co, device=dev) + 0.5, torch.rand(8, ci, device=dev) + 0.5
    bn[2], bn[6], bn[7], mbn[7] = 0.1, 0.0, 0.0, 0.0
    bn[7, 1] = 0.05  # TTK_AUX_DY_BOUND
    a_bound = torch.tensor([8.0], device=dev)
    dy = torch.empty_like(g)
    nb = L.conv_wgrad_partial_bytes(B, hw, hw, ci, co, k, s)
    scr = torch.empty(nb // 4, device=dev) if nb and not os.environ.get("ATOMIC") else None  # slice-wise weight gradient (ATOMIC=1: fp32 atomics)
    yp = p(y) if os.environ.get("TTK_GEMM") == "bf16x3" or os.environ.get("ONLOAD") else None  # None: the gradients read the materialised dy
    wf, wb = torch.empty(3, k * k, co, ci, dtype=torch.int16, device=dev), torch.empty(3, k * k, ci, co, dtype=torch.int16, device=dev)
    L.call("ttk_conv_weight_repack", p(w), p(wf), p(wb), co, ci, k, k)
    out, gin, dw = torch.empty(B, ho, ho, co, device=dev), torch.empty(B, hw, hw, ci, device=dev), torch.zeros(co, ci, k, k, device=dev)
    part = torch.empty(max(L.partial_rows_gemm(M), L.partial_rows_gemm(B * hw * hw)) * 2 * max(ci, co), device=dev)
    calls = {
        "fwd": lambda: L.call("ttk_conv_fwd", p(a), p(a_bound), p(wf), p(out), p(part), None, B, hw, hw, ci, co, k, k, s, pad),
        "dy": lambda: L.call("ttk_bn_bwd_apply", p(g), p(y), p(bn), p(dy), M, co),
        "dgrad": lambda: L.call("ttk_conv_bwd_data", p(dy), yp, p(bn), p(wb), None, None, p(gin), None, B, hw, hw, ci, co, k, k, s, pad),
        "dgrad_m": lambda: L.call("ttk_conv_bwd_data", p(dy), yp, p(bn), p(wb), p(a), p(mbn), p(gin), p(part), B, hw, hw, ci, co, k, k, s, pad),
        "wgrad": lambda: L.call("ttk_conv_bwd_weight", p(dy), yp, p(bn), p(a), p(a_bound), p(dw), p(scr), B, hw, hw, ci, co, k, k, s, pad),
    }
    line = f"{name:4s} M={M:7d} K={k * k * ci:5d} N={co:4d} "
    for kk, fn in calls.items():
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(IT):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / IT
        line += f"| {kk} {us:7.1f} us {2 * M * k * k * ci * co / us / 1e6:6.1f} TF "
        tot[kk] += us * mult
    print(line, flush=True)
print("per-step totals (us; dgrad and dgrad_m are alternatives):", {k: round(v) for k, v in tot.items()})
