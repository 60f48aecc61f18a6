"""Time the forward / data-gradient GEMM of a K=N=C layer against M (row tiles of 128): shows round quantisation."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda"
for R in [16, 32, 64, 128, 192, 256, 288, 320, 324, 352, 384, 448, 512, 640, 768, 1024, 2048]:
    M = R * 128
    ydw, y, g = torch.randn(M, C, device=dev), torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)
    w = torch.randn(C, C, device=dev) * 0.05
    bn = torch.rand(8, C, device=dev) + 0.5
    out = torch.empty(M, C, device=dev)
    wq = torch.empty(3 * 1024 * 1024, dtype=torch.int16, device=dev)
    part = torch.empty(L.partial_rows_gemm(M) * 2 * C, device=dev)
    calls = {"fwd": lambda: L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn), p(w), p(out), p(part), M, C, C, p(wq)),
             "dgrad": lambda: L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bn), p(w), p(ydw), p(bn), p(out), p(part), M, C, C, p(wq))}
    line = f"R={R:5d} tiles={R*C//128:6d} ({R*C/128/256:6.2f}/CU) "
    for k, fn in calls.items():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        line += f"| {k} {us:7.1f} us {2*M*C*C/us/1e6:6.1f} TF "
    print(line)
