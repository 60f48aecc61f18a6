"""Runs the forward GEMM of one layer shape on a TTK_EXP=9 build and prints the main-loop cycles / real time of tile 0."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
for (M, ci, co) in [(41472, 512, 512), (12800, 1024, 1024), (147968, 256, 256)]:
    dev = "cuda"
    ydw, w = torch.randn(M, ci, device=dev), torch.randn(co, ci, device=dev) * (2.0 / co) ** 0.5
    bn = torch.rand(8, ci, device=dev) + 0.5
    bn[7] = 0.0; bn[7, 0] = 12.0
    out = torch.empty(M, co, device=dev)
    prep = torch.empty(L.pwconv_prepared_bytes(ci, co), dtype=torch.uint8, device=dev)
    L.pwconv_prepare_weights([w], [prep])
    rows = L.partial_rows_gemm(M)
    part = torch.zeros(rows * 2 * co + 64, device=dev)
    for _ in range(200):  # sustained load: the clock settles
        L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn), None, p(out), p(part), None, M, ci, co, p(prep), 0)
    torch.cuda.synchronize()
    c, t, n, bw, pl, tb, ta = part[rows * 2 * co: rows * 2 * co + 7].tolist()
    print(f"M={M} K={ci} N={co}: main loop {c:.0f} cycles / {n:.0f} steps = {c / n:.0f} cycles per k32 step; {t * 10:.0f} ns -> clock {c / (t * 10):.3f} GHz; "
          f"per step: consumer barrier wait {bw / n:.0f}; producer loop {pl / n:.0f} = B part {tb / n:.0f} + A part {ta / n:.0f} + barrier wait {(pl - tb - ta) / n:.0f}")
