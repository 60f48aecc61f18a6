"""Fixed cost per tile of pw_split_k: forward / data-gradient time vs K at M = 32768 (256 M-tiles), N = 256 -> exactly one
round of 256 tiles per launch, so time = prologue + K/32 steps + epilogue of ONE tile."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
M, N = 32768, 256
for K in (128, 256, 512, 1024):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05
    bn = torch.rand(8, K, device="cuda") + 0.5; bno = torch.rand(8, N, device="cuda") + 0.5
    y = torch.empty(M, N, device="cuda"); part = torch.zeros(L.partial_rows_gemm(M), 2, max(N, K), device="cuda")
    prep = torch.empty(L.pwconv_prepared_bytes(K, N), dtype=torch.uint8, device="cuda")
    L.pwconv_prepare_weights([w], [prep])
    # data gradient of a layer with Cin = N... use the forward only plus the transposed problem for dgrad timing
    f = lambda: L.call("ttk_pwconv1x1_fwd", p(x), p(bn), None, p(y), p(part), M, K, N, p(prep))
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    print(f"K={K:5d}  steps={K // 32:3d}  {s.elapsed_time(e) / 20 * 1e3:7.1f} us per one-round launch")
