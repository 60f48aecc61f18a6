"""GPU micro-benchmark of the three pointwise-conv entry points on the 13 layer shapes of the default backbone at batch B
(operands prepared once, as the training step does).  python tools/bench_gemm.py [B] [iters]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H  # noqa: E402

L, p = H.lib(), H.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
BF = int(os.environ.get("BF16", "0"))  # BF16=1: bf16 activation storage
ADT = torch.bfloat16 if BF else torch.float32
shapes = [("dw2_1", 65, 32, 64, 1), ("dw2_2", 33, 64, 128, 1), ("dw3_1", 33, 128, 128, 1), ("dw3_2", 17, 128, 256, 1), ("dw4_1", 17, 256, 256, 1),
          ("dw4_2", 9, 256, 512, 1), ("dw5_x", 9, 512, 512, 5), ("dw5_6", 5, 512, 1024, 1), ("dw6", 5, 1024, 1024, 1)]
if os.environ.get("LAYERS"):  # LAYERS=dw5_x,dw6: only these
    shapes = [s for s in shapes if s[0] in os.environ["LAYERS"].split(",")]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
dev = "cuda"
for name, hw, ci, co, mult in shapes:
    M = B * hw * hw
    ydw, y, g = torch.randn(M, ci, device=dev).to(ADT), torch.randn(M, co, device=dev).to(ADT), (torch.randn(M, co, device=dev) * 1e-3).to(ADT)
    w = torch.randn(co, ci, device=dev) * (2.0 / co) ** 0.5
    bn_dw, bn_pw = torch.rand(8, ci, device=dev) + 0.5, torch.rand(8, co, device=dev) + 0.5
    bn_dw[2], bn_pw[2], bn_pw[6] = 0.1, 0.1, 0.0  # means
    bn_dw[7], bn_pw[7] = 0.0, 0.0
    bn_dw[7, 0], bn_pw[7, 1] = 12.0, 0.05  # TTK_AUX_ACT_BOUND, TTK_AUX_DY_BOUND (generous for this data)
    out, gdw, dW = torch.empty(M, co, device=dev, dtype=ADT), torch.empty(M, ci, device=dev, dtype=ADT), torch.zeros(co, ci, device=dev)
    prep = torch.empty(L.pwconv_prepared_bytes(ci, co), dtype=torch.uint8, device=dev)
    L.pwconv_prepare_weights([w], [prep])
    nb = L.pwconv_wgrad_partial_bytes(M, ci, co) if os.environ.get("PARTIAL") else L.pwconv_wgrad_scratch_bytes(M, ci, co)  # PARTIAL=1: slice partials + fixed-order fold everywhere
    scr = torch.empty(nb // 4, device=dev) if nb else None
    part = torch.empty(max(L.partial_rows_gemm(M, ci, co), L.partial_rows_gemm(M, co, ci, True)) * 2 * max(ci, co), device=dev)
    calls = {
        "fwd": lambda: L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), None, p(out), p(part), None, M, ci, co, p(prep), BF),
        "dgrad": lambda: L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bn_pw), None, p(ydw), p(bn_dw), p(gdw), p(part), M, ci, co, p(prep), BF),
        "wgrad": lambda: L.call("ttk_pwconv1x1_bwd_weight", p(g), p(y), p(bn_pw), p(ydw), p(bn_dw), p(dW), p(scr), M, ci, co, BF),
    }
    if L.cdll.ttk_pwconv1x1_bwd_fused_rows(M, ci, co) and not BF:  # the first two layers: both gradients in one kernel
        fpart = torch.empty(L.cdll.ttk_pwconv1x1_bwd_fused_rows(M, ci, co) * 2 * ci, device=dev)
        calls["fused"] = lambda: L.call("ttk_pwconv1x1_bwd_fused", p(g), p(y), p(bn_pw), p(w), p(prep), p(ydw), p(bn_dw), p(gdw), p(dW), None, p(fpart), M, ci, co)
    line = f"{name:6s} M={M:8d} K={ci:4d} N={co:4d} "
    if os.environ.get("KINDS"):  # KINDS=fwd,dgrad
        calls = {k: v for k, v in calls.items() if k in os.environ["KINDS"].split(",")}
    for k, fn in calls.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(IT):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / IT
        line += f"| {k} {us:7.1f} us {2 * M * ci * co / us / 1e6:6.1f} TF "
        tot[k] = tot.get(k, 0.0) + us * mult
    print(line, flush=True)
print("per-step totals (us):", {k: round(v) for k, v in tot.items()}, "sum", round(sum(tot.values())))
