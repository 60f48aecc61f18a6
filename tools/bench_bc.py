"""GPU micro-benchmark of the bf16-compute entry points (csrc/bc_*.hip) on the layer shapes of the default backbone at batch B: the two
depthwise kernels and the three pointwise products, each timed alone with HIP events over `iters` back-to-back launches.
    python tools/bench_bc.py [B] [iters] [dw|pw|all]          (TTK_LIB=... selects an experiment build)"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H  # noqa: E402

L, p = H.lib(), H.ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 10
WHAT = sys.argv[3] if len(sys.argv) > 3 else "all"
dev, BF = "cuda", torch.bfloat16
# (name, input size, cin, cout, stride, residual block, repetitions per step)
blocks = [("dw2_1", 65, 32, 64, 1, False, 1), ("dw2_2", 65, 64, 128, 2, False, 1), ("dw3_1", 33, 128, 128, 1, True, 1), ("dw3_2", 33, 128, 256, 2, False, 1),
          ("dw4_1", 17, 256, 256, 1, True, 1), ("dw4_2", 17, 256, 512, 2, False, 1), ("dw5_x", 9, 512, 512, 1, True, 5), ("dw5_6", 9, 512, 1024, 2, False, 1),
          ("dw6", 5, 1024, 1024, 1, True, 1)]


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(IT):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / IT


tot = {}
prev_skip = False  # the producer of this block's input was a residual block (its block input is added before the ReLU)
for name, h, ci, co, s, res, mult in blocks:
    ho = (h - 1) // s + 1
    n_in, n_out = B * h * h * ci, B * ho * ho * ci
    M = B * ho * ho
    rnd = lambda *sh: torch.randn(*sh, device=dev).to(BF)
    bn = lambda C: torch.cat([torch.rand(1, C, device=dev) + 0.5, torch.randn(2, C, device=dev) * 0.2, torch.rand(2, C, device=dev) + 0.5, torch.randn(2, C, device=dev) * 0.05,
                              torch.zeros(1, C, device=dev)]).contiguous()
    line = f"{name:6s} {h:2d}x{h:<2d} {ci:4d}->{co:<4d} s{s} "
    if WHAT in ("dw", "all"):
        yprev, skp, ydw, gdw = rnd(B, h, h, ci), (rnd(B, h, h, ci).abs() if prev_skip else None), rnd(B, ho, ho, ci), rnd(B, ho, ho, ci) * 0.01
        a_out = torch.empty_like(yprev) if res else None
        sg = rnd(B, h, h, ci) * 0.01 if res else None
        w, bnp, bnd = torch.randn(ci, 1, 3, 3, device=dev) * 0.3, bn(ci), bn(ci)
        gprev, dwg = torch.empty_like(yprev), torch.zeros(ci, 1, 3, 3, device=dev)
        part = torch.empty(max(L.cdll.ttk_bc_partial_rows_dw(B, h, h, ci, s, 0), L.cdll.ttk_bc_partial_rows_dw(B, h, h, ci, s, 1)) * 2 * ci, device=dev)
        us = timed(lambda: L.call("ttk_bc_dw_fwd", p(yprev), p(bnp), p(skp), p(a_out), p(w), p(ydw), p(part), None, B, h, h, ci, s))
        by = 2 * (n_in * (1 + (skp is not None) + res) + n_out)
        line += f"| dw fwd {us:6.1f} us {by / us / 1e3:5.0f} GB/s "
        tot["dw_fwd"] = tot.get("dw_fwd", 0.0) + us * mult
        rows_scr = torch.empty(L.cdll.ttk_bc_partial_rows_dw(B, h, h, ci, s, 1) * 9 * ci, device=dev)  # the product's form: workgroup rows + fold
        us = timed(lambda: L.call("ttk_bc_dw_bwd_data", p(gdw), p(ydw), p(bnd), p(w), p(sg), p(yprev), p(bnp), p(skp), p(a_out), p(gprev), p(part), p(dwg), 1, p(rows_scr),
                                   B, h, h, ci, s))
        by = 2 * (2 * n_out + n_in * (2 + res + (res or skp is not None)))
        line += f"| dw bwd {us:6.1f} us {by / us / 1e3:5.0f} GB/s "
        tot["dw_bwd"] = tot.get("dw_bwd", 0.0) + us * mult
    if WHAT in ("pw", "all"):
        ydw2, y, g = rnd(M, ci), rnd(M, co), rnd(M, co) * 0.01
        wp = torch.randn(co, ci, 1, 1, device=dev) * (2.0 / co) ** 0.5
        bnd, bnq = bn(ci), bn(co)
        prep = torch.empty(L.cdll.ttk_bc_prepared_bytes(ci, co), dtype=torch.uint8, device=dev)
        L.bc_prepare_weights([wp], [prep])
        out, gd, dW = torch.empty(M, co, device=dev, dtype=BF), torch.empty(M, ci, device=dev, dtype=BF), torch.zeros(co, ci, device=dev)
        part = torch.empty(max(L.cdll.ttk_bc_partial_rows_pw(M, ci, co), L.cdll.ttk_bc_partial_rows_pw(M, co, ci)) * 2 * max(ci, co), device=dev)
        scr = torch.empty(L.cdll.ttk_bc_pw_wgrad_scratch_bytes(M, ci, co) // 4, device=dev)
        for k, fn, by in (("fwd", lambda: L.call("ttk_bc_pw_fwd", p(ydw2), p(bnd), p(prep), p(out), p(part), None, M, ci, co), 2 * M * (ci + co)),
                          ("dgrad", lambda: L.call("ttk_bc_pw_bwd_data", p(g), p(y), p(bnq), p(prep), p(ydw2), p(bnd), p(gd), p(part), M, ci, co), 2 * M * 2 * (ci + co)),
                          ("wgrad", lambda: L.call("ttk_bc_pw_bwd_weight", p(g), p(y), p(bnq), p(ydw2), p(bnd), p(dW), p(scr), M, ci, co), 2 * M * (2 * co + ci))):
            us = timed(fn)
            line += f"| pw {k} {us:6.1f} us {by / us / 1e3:5.0f} GB/s {2 * M * ci * co / us / 1e6:5.0f} TF "
            tot["pw_" + k] = tot.get("pw_" + k, 0.0) + us * mult
        fr = L.cdll.ttk_bc_pw_bwd_fused_rows(M, ci, co)
        if fr > 0:  # what the step runs INSTEAD of dgrad + wgrad on the early layers
            scr2, part2 = torch.empty(L.cdll.ttk_bc_pw_bwd_fused_scratch_bytes(M, ci, co) // 4, device=dev), torch.empty(fr * 2 * ci, device=dev)
            us = timed(lambda: L.call("ttk_bc_pw_bwd_fused", p(g), p(y), p(bnq), p(prep), p(ydw2), p(bnd), p(gd), p(dW), p(scr2), p(part2), M, ci, co))
            line += f"| fused bwd {us:6.1f} us {2 * M * 2 * (ci + co) / us / 1e3:5.0f} GB/s "
            tot["pw_fused_bwd"] = tot.get("pw_fused_bwd", 0.0) + us * mult
    print(line, flush=True)
    prev_skip = res
print("per-step totals (us):", {k: round(v) for k, v in tot.items()}, "sum", round(sum(tot.values())))
