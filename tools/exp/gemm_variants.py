"""Times ttk_pwconv1x1_fwd / _bwd_data of the timing-only variants built by gemm_variants.sh (results are wrong for N>0)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H

HERE = os.path.dirname(os.path.abspath(__file__))
shapes = [(41472, 512, 512), (12800, 1024, 1024), (147968, 256, 256)]
names = {0: "baseline", 7: "A rows mod 1024 (L2-resident A)", 8: "exp8", 9: "exp9"}
variants = [int(a) for a in sys.argv[1:]] or sorted(names)
for v in variants:
    H.LIB_PATH = os.path.join(HERE, "_build", f"libttk_exp{v}.so")
    H._lib = None
    L, p = H.lib(), H.ptr
    line = f"{v} {names.get(v, '?'):18s}"
    for M, Cin, Cout in shapes:
        x = torch.randn(M, Cin, device="cuda"); w = torch.randn(Cout, Cin, device="cuda") * 0.05
        bn = torch.rand(8, Cin, device="cuda") + 0.5; bno = torch.rand(8, Cout, device="cuda") + 0.5
        y = torch.empty(M, Cout, device="cuda"); g = torch.randn(M, Cout, device="cuda"); gd = torch.empty(M, Cin, device="cuda")
        part = torch.empty(L.partial_rows_gemm(M) * 2 * max(Cin, Cout), device="cuda")
        prep = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device="cuda")
        L.pwconv_prepare_weights([w], [prep])
        def fwd(): L.call("ttk_pwconv1x1_fwd", p(x), p(bn), None, p(y), p(part), M, Cin, Cout, p(prep))
        def dgr(): L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bno), None, p(x), p(bn), p(gd), p(part), M, Cin, Cout, p(prep))
        for f in (fwd, dgr):
            for _ in range(3): f()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): f()
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1e3
            line += f"  {us:7.1f}us {2.0 * M * Cin * Cout / us / 1e6:6.1f}TF"
    print(line, flush=True)
