"""pw_big_fwd_k (TTK_GEMM=big) against pw_split_k: run once per mode, the second run compares with the first's dump."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
L, p = H.lib(), H.ptr
mode = os.environ.get("TTK_GEMM", "split")
dump = "/tmp/big_check.pt"
res = {}
for M, Cin, Cout in ((41472, 512, 512), (12800, 1024, 1024), (147968, 256, 256), (41472, 256, 512), (12800, 512, 1024), (1000, 128, 256), (147968, 128, 256)):
    g = torch.Generator().manual_seed(M + Cin)
    x = torch.randn(M, Cin, generator=g).cuda(); w = (torch.randn(Cout, Cin, generator=g) * 0.05).cuda()
    bn = (torch.rand(8, Cin, generator=g) + 0.5).cuda()
    y = torch.empty(M, Cout, device="cuda")
    rows = L.partial_rows_gemm(M)
    part = torch.zeros(rows, 2, Cout, device="cuda")
    prep = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device="cuda")
    L.pwconv_prepare_weights([w], [prep])
    f = lambda: L.call("ttk_pwconv1x1_fwd", p(x), p(bn), None, p(y), p(part), M, Cin, Cout, p(prep))
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    res[(M, Cin, Cout)] = (y.cpu(), part.cpu())
    line = f"{mode:6s} M={M:7d} K={Cin:5d} N={Cout:5d}  {us:7.1f} us  {2.0 * M * Cin * Cout / us / 1e6:6.1f} TF"
    if os.path.exists(dump) and mode != "split":
        ref = torch.load(dump)[(M, Cin, Cout)]
        dy = (ref[0] - res[(M, Cin, Cout)][0]).abs().max().item()
        dp = (ref[1] - res[(M, Cin, Cout)][1]).abs().max().item() / max(ref[1].abs().max().item(), 1e-30)
        line += f"   max|dy| {dy:.3e}  rel d(part) {dp:.2e}"
    print(line, flush=True)
if mode == "split":
    torch.save(res, dump)
