"""Where does a pointwise GEMM entry point differ from float64?  python tools/debug/x_check.py M Cin Cout [loose] [reps]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import trackertraincode._hip as H
from test_pwconv_gpu import _bn_block, BN_SCALE, BN_BETA, BN_MEAN, BN_GA, BN_GB, BN_GMEAN, BN_AUX, AUX_ACT_BOUND, AUX_DY_BOUND
M, Cin, Cout = (int(x) for x in sys.argv[1:4]); loose = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
L, p = H.lib(), H.ptr
rng = np.random.default_rng(M + Cin + Cout)
ydw = rng.normal(0, 1, (M, Cin)).astype(np.float32); w = (rng.normal(0, 1, (Cout, Cin)) * np.sqrt(2.0 / Cout)).astype(np.float32)
bn_dw, bn_pw = _bn_block(Cin, rng), _bn_block(Cout, rng)
t = lambda a: torch.from_numpy(a).cuda()
a32 = np.maximum(bn_dw[BN_SCALE] * (ydw - bn_dw[BN_MEAN]) + bn_dw[BN_BETA], 0).astype(np.float32)
bn_dw[BN_AUX, AUX_ACT_BOUND] = np.abs(a32).max() * loose
y64 = a32.astype(np.float64) @ w.astype(np.float64).T
g = rng.normal(0, 1, (M, Cout)).astype(np.float32)
d_ydw, d_w, d_bn = H.to_blocks(t(ydw)), t(w), t(bn_dw)
wq = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device="cuda")
rows, rows_b = L.partial_rows_gemm(M, Cin, Cout), L.partial_rows_gemm(M, Cout, Cin, True)
print("tile rows fwd/dgrad", L.cdll.ttk_pwconv_tile_rows(M, Cin, Cout, 0), L.cdll.ttk_pwconv_tile_rows(M, Cout, Cin, 1), "partial rows", rows, rows_b)
def report(name, got, ref):
    err = np.abs(got - ref); tol = 1e-5 * np.abs(ref).max()
    bad = np.argwhere(err > tol)
    print(f"{name}: rel {np.linalg.norm(got - ref) / np.linalg.norm(ref):.3e}, bad elements {len(bad)} of {got.size}", end="")
    if len(bad):
        px, ch = np.unique(bad[:, 0]), np.unique(bad[:, 1])
        print(f"; pixels {px[:12]}{'...' if len(px) > 12 else ''} ({len(px)}), channels {ch[:12]}{'...' if len(ch) > 12 else ''} ({len(ch)}); worst {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} ref {ref[np.unravel_index(err.argmax(), err.shape)]:.4f}", end="")
    print()
for r in range(reps):
    y = torch.empty(M, Cout, device="cuda"); part = torch.zeros(rows, 2, Cout, device="cuda")
    L.call("ttk_pwconv1x1_fwd", p(d_ydw), p(d_bn), p(d_w), p(y), p(part), None, M, Cin, Cout, p(wq), 0)
    torch.cuda.synchronize()
    yv = H.from_blocks(y).cpu().numpy()
    report("fwd  ", yv.astype(np.float64), y64)
    dy32 = (bn_pw[BN_GA] * (g - bn_pw[BN_GMEAN]) + bn_pw[BN_GB] * (yv - bn_pw[BN_MEAN])).astype(np.float32)
    bn_pw[BN_AUX, AUX_DY_BOUND] = np.abs(dy32).max() * loose
    pre = bn_dw[BN_SCALE] * (ydw - bn_dw[BN_MEAN]) + bn_dw[BN_BETA]; mask = pre > 0; safe = np.abs(pre) > 1e-4
    gd64 = (dy32.astype(np.float64) @ w.astype(np.float64)) * mask
    wt = torch.from_numpy(np.ascontiguousarray(w.T)).cuda(); g_dw = torch.empty(M, Cin, device="cuda"); part2 = torch.zeros(rows_b, 2, Cin, device="cuda")
    d_g, d_bnpw = H.to_blocks(t(g)), t(bn_pw)
    L.call("ttk_pwconv1x1_bwd_data", p(d_g), p(y), p(d_bnpw), p(wt), p(d_ydw), p(d_bn), p(g_dw), p(part2), M, Cin, Cout, p(wq), 0)
    torch.cuda.synchronize()
    report("dgrad", H.from_blocks(g_dw).cpu().numpy().astype(np.float64) * safe, gd64 * safe)
