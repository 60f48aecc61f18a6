"""Timing-only variants of pw_big_fwd_k (gemm_variants.sh 31..34; results wrong): 31 no A conversion, 32 no MFMA phase,
33 no B transfers, 34 no A loads."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import trackertraincode._hip as H
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["TTK_GEMM"] = "big"
names = {0: "product lib", 31: "no A conversion", 32: "no MFMA phase", 33: "no B transfers", 34: "no A loads", 42: "interleave 2 VALU/MFMA", 44: "interleave 4 VALU/MFMA"}
for v in [int(a) for a in sys.argv[1:]] or [0, 31, 32, 33, 34]:
    H.LIB_PATH = os.path.join(HERE, "_build", f"libttk_exp{v}.so") if v else H.LIB_PATH
    if v: H._lib = None
    L, p = H.lib(), H.ptr
    line = f"{v:3d} {names[v]:18s}"
    for M, Cin, Cout in ((65536, 512, 512), (65536, 1024, 1024)):  # 512 / 1024 tiles: exactly 2 / 4 rounds
        x = torch.randn(M, Cin, device="cuda"); w = torch.randn(Cout, Cin, device="cuda") * 0.05
        bn = torch.rand(8, Cin, device="cuda") + 0.5
        y = torch.empty(M, Cout, device="cuda"); part = torch.zeros(L.partial_rows_gemm(M), 2, Cout, device="cuda")
        prep = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device="cuda")
        L.pwconv_prepare_weights([w], [prep])
        f = lambda: L.call("ttk_pwconv1x1_fwd", p(x), p(bn), None, p(y), p(part), M, Cin, Cout, p(prep))
        for _ in range(3): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 10 * 1e3
        line += f"  {us:7.1f} us {2.0 * M * Cin * Cout / us / 1e6:6.1f} TF"
    print(line, flush=True)
