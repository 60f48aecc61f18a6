"""Range stress of the fp16-split GEMMs through the real C-ABI entry points (ttk_pwconv1x1_*, ttk_conv_*).

The split kernels (csrc/pwconv_f16.hip) scale every operand tensor by a power of two S taken from an upper BOUND of its
magnitude (row TTK_BN_AUX), S * bound in [2^14, 2^15), and cut x S into two fp16 pieces.  Documented behaviour
(DESIGN.md 4.1): an element down to 2^-17 of the (power-of-two) bound keeps 22 bits; a smaller one keeps an ABSOLUTE error of
2^-25 / S = 2^-40 of the bound; below 2^-40 of the bound it is gone.  The training step's own bounds are 1-100x loose and its
operands span a few decades; here the operands are heavy-tailed (log-normal, sigma = 4: fourteen decades) and the bounds are
2^0 ... 2^12 loose, and the kernels must follow that curve:

 * the result is as close to the float64 product as an fp32 multiply-add chain is, plus the floor term
   sum_k (floor_a |b| + |a| floor_b), floor = 2^-25 / S for elements under 2^-17 of the bound - at every looseness;
 * probe rows whose elements ALL sit 2^-10 ... 2^-36 below the bound come out non-zero with the relative error the floor
   predicts: nothing inside the supported range (>= 2^-39 of the bound) is flushed to zero.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BN_SCALE, BN_BETA, BN_MEAN, BN_RSTD, BN_GA, BN_GB, BN_GMEAN, BN_AUX = range(8)
AUX_ACT_BOUND, AUX_DY_BOUND, AUX_GMAX = range(3)
PROBES = (10, 20, 30, 36)  # probe row i holds elements of magnitude 2^-PROBES[i] x the bound


def _pow2_scale(bound):
    return 2.0 ** (14 - int(np.floor(np.log2(float(bound)))))


def _floor(x, bound):
    """absolute error floor of every element of x under the split with `bound`: 0 where all 22 bits are kept"""
    S = _pow2_scale(bound)
    return np.where(np.abs(x) * S >= 2.0 ** -3, 0.0, 2.0 ** -25 / S)


def _chain32(a, b_t):
    acc = np.zeros((a.shape[0], b_t.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc += a[:, k:k + 1] * b_t[k:k + 1, :]
    return acc


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _lognormal(rng, shape, sigma, signed):
    x = np.exp(sigma * rng.standard_normal(shape))
    if signed:
        x *= np.where(rng.random(shape) < 0.5, -1.0, 1.0)
    return x.astype(np.float32)


def _with_probes(x, bound_true, signed, rng):
    """rows 0..len(PROBES)-1 become probe rows: every element 2^-d of the tensor's true maximum (random signs if signed)"""
    x = x.copy()
    for i, d in enumerate(PROBES):
        row = np.full(x.shape[1], bound_true * 2.0 ** -d, np.float32) * rng.uniform(0.5, 1.0, x.shape[1]).astype(np.float32)
        if signed:
            row *= np.where(rng.random(x.shape[1]) < 0.5, -1.0, 1.0).astype(np.float32)
        x[i] = row
    return x


def _check(name, out, a, b_t, bound_a, bound_b, loose, probe_rows):
    """out[M,N] = a[M,K] @ b_t[K,N] computed by the kernel under bounds (bound_a, bound_b)."""
    ref = a.astype(np.float64) @ b_t.astype(np.float64)
    assert np.isfinite(out).all(), name
    e_hip, e_f32 = _rel(out, ref), _rel(_chain32(a, b_t), ref)
    fa, fb = _floor(a, bound_a), _floor(b_t, bound_b)
    floor_term = fa @ np.abs(b_t).astype(np.float64) + np.abs(a).astype(np.float64) @ fb
    e_floor = float(np.linalg.norm(floor_term) / np.linalg.norm(ref))
    print(f"{name} loose 2^{int(np.log2(loose))}: hip {e_hip:.2e}  fp32 chain {e_f32:.2e}  floor term {e_floor:.2e}")
    assert e_hip <= 1.5 * e_f32 + e_floor + 3e-7, (name, loose, e_hip, e_f32, e_floor)  # (h + l reproduces x S to 2^-23 per operand)
    for i in probe_rows:  # nothing flushes to zero inside the supported range
        r_ref, r_out = ref[i], np.asarray(out[i], np.float64)
        tol = floor_term[i] + 4e-7 * (np.abs(a[i]).astype(np.float64) @ np.abs(b_t).astype(np.float64))
        assert np.all(np.abs(r_out - r_ref) <= tol), (name, loose, i, float(np.abs(r_out - r_ref).max()), float(tol.min()))
        if floor_term[i].max() < 0.25 * np.abs(r_ref).max():
            assert np.abs(r_out).max() > 0.5 * np.abs(r_ref).max(), (name, loose, i, "flushed to zero")


@pytest.mark.parametrize("loose", [1.0, 2.0 ** 4, 2.0 ** 8, 2.0 ** 12])
@pytest.mark.parametrize("M,Cin,Cout,wsigma", [(2049, 512, 512, 0.0), (1300, 256, 128, 2.0)])
def test_pwconv_entry_points_under_heavy_tails_and_loose_bounds(M, Cin, Cout, wsigma, loose):
    import trackertraincode._hip as H
    L, p = H.lib(), H.ptr
    rng = np.random.default_rng(int(np.log2(loose)) + M)
    dev = "cuda"
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    # operands: A = relu(1 * (ydw - 0) + 0) = ydw > 0 log-normal; weights gaussian or log-normal; dy = 1 * (g - 0) + 0 * (y - 0) = g signed log-normal
    ydw = _lognormal(rng, (M, Cin), 4.0, signed=False)
    ydw = _with_probes(ydw, float(ydw.max()), False, rng)
    w = (rng.standard_normal((Cout, Cin)) * np.sqrt(2.0 / Cout)).astype(np.float32) if wsigma == 0 else _lognormal(rng, (Cout, Cin), wsigma, True) * np.float32(0.01)
    g = _lognormal(rng, (M, Cout), 4.0, signed=True)
    g = _with_probes(g, float(np.abs(g).max()), True, rng)
    bn_dw, bn_pw = np.zeros((8, Cin), np.float32), np.zeros((8, Cout), np.float32)
    bn_dw[BN_SCALE], bn_dw[BN_RSTD], bn_pw[BN_GA], bn_pw[BN_SCALE], bn_pw[BN_RSTD] = 1.0, 1.0, 1.0, 1.0, 1.0
    bound_a, bound_g, bound_w = float(ydw.max()) * loose, float(np.abs(g).max()) * loose, float(np.abs(w).max())  # (the weights' bound is their measured maximum)
    bn_dw[BN_AUX, AUX_ACT_BOUND], bn_pw[BN_AUX, AUX_DY_BOUND] = bound_a, bound_g
    d_ydw, d_w, d_bndw, d_bnpw, d_g = H.to_blocks(t(ydw)), t(w), t(bn_dw), t(bn_pw), H.to_blocks(t(g))  # activations: channel blocks (include/ttk.h)
    rows, rows_b = L.partial_rows_gemm(M, Cin, Cout), L.partial_rows_gemm(M, Cout, Cin, True)
    probes = range(len(PROBES))

    y = torch.empty(M, Cout, device=dev)
    part = torch.empty(rows, 2, Cout, device=dev)
    wq = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
    L.call("ttk_pwconv1x1_fwd", p(d_ydw), p(d_bndw), p(d_w), p(y), p(part), None, M, Cin, Cout, p(wq), 0)
    torch.cuda.synchronize()
    _check("fwd", H.from_blocks(y).cpu().numpy(), ydw, np.ascontiguousarray(w.T), bound_a, bound_w, loose, probes)

    wt = t(w.T)
    g_dw = torch.empty(M, Cin, device=dev)
    part2 = torch.empty(rows_b, 2, Cin, device=dev)
    y0 = torch.zeros(M, Cout, device=dev)  # gb = 0: the conv output does not enter dy
    L.call("ttk_pwconv1x1_bwd_data", p(d_g), p(y0), p(d_bnpw), p(wt), p(d_ydw), p(d_bndw), p(g_dw), p(part2), M, Cin, Cout, p(wq), 0)
    torch.cuda.synchronize()
    _check("dgrad", H.from_blocks(g_dw).cpu().numpy(), g, w, bound_g, bound_w, loose, probes)  # (ydw > 0 everywhere: the ReLU mask is all ones)

    dW = torch.zeros(Cout, Cin, device=dev)
    L.call("ttk_pwconv1x1_bwd_weight", p(d_g), p(y0), p(d_bnpw), p(d_ydw), p(d_bndw), p(dW), None, M, Cin, Cout, 0)
    torch.cuda.synchronize()
    _check("wgrad", dW.cpu().numpy(), np.ascontiguousarray(g.T), ydw, bound_g, bound_a, loose, ())


@pytest.mark.parametrize("loose", [1.0, 2.0 ** 6, 2.0 ** 12])
def test_conv_entry_points_under_heavy_tails_and_loose_bounds(loose):
    """The implicit-GEMM convolutions of the ResNet18 variant (3x3, 128 -> 128 at 17x17): forward, data gradient and
    weight gradient against float64 torch convolutions, with the floor term evaluated through the same convolutions."""
    import trackertraincode._hip as Hh
    L, p = Hh.lib(), Hh.ptr
    B, H, Cin, Cout, k, stride, pad = 2, 17, 128, 128, 3, 1, 1
    rng = np.random.default_rng(int(np.log2(loose)) + 3)
    dev = "cuda"
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    a = _lognormal(rng, (B, H, H, Cin), 4.0, signed=False)
    w = (rng.standard_normal((Cout, Cin, k, k)) * np.sqrt(2.0 / (k * k * Cout))).astype(np.float32)
    g = _lognormal(rng, (B, H, H, Cout), 4.0, signed=True)
    bound_a, bound_g, bound_w = float(a.max()) * loose, float(np.abs(g).max()) * loose, float(np.abs(w).max())
    bn = np.zeros((8, Cout), np.float32)
    bn[BN_GA], bn[BN_SCALE], bn[BN_RSTD] = 1.0, 1.0, 1.0
    bn[BN_AUX, AUX_DY_BOUND] = bound_g
    d_a, d_w, d_g, d_bn = t(a), t(w), t(g), t(bn)
    a_bound = torch.tensor([bound_a], device=dev)
    w_f = torch.empty(3, k * k, Cout, Cin, dtype=torch.int16, device=dev)
    w_b = torch.empty(3, k * k, Cin, Cout, dtype=torch.int16, device=dev)
    L.call("ttk_conv_weight_repack", p(d_w), p(w_f), p(w_b), Cout, Cin, k, k)
    nchw = lambda x: torch.from_numpy(np.ascontiguousarray(x)).double().permute(0, 3, 1, 2)
    nhwc = lambda x: x.permute(0, 2, 3, 1).numpy()
    a64, w64, g64 = nchw(a), torch.from_numpy(w).double(), nchw(g)
    a32, w32, g32 = a64.float(), w64.float(), g64.float()
    fa, fw, fg = nchw(_floor(a, bound_a)), torch.from_numpy(_floor(w, bound_w)), nchw(_floor(g, bound_g))

    def judge(name, out, ref, ref32, floor_term):
        e_hip, e_f32 = _rel(out, ref), _rel(ref32, ref)
        e_floor = float(np.linalg.norm(floor_term) / np.linalg.norm(ref))
        print(f"conv {name} loose 2^{int(np.log2(loose))}: hip {e_hip:.2e}  fp32 torch {e_f32:.2e}  floor term {e_floor:.2e}")
        assert np.isfinite(out).all()
        assert e_hip <= 1.5 * e_f32 + e_floor + 1.5e-6, (name, loose, e_hip, e_f32, e_floor)

    M = B * H * H
    y = torch.empty(B, H, H, Cout, device=dev)
    part = torch.empty(L.partial_rows_gemm(M), 2, Cout, device=dev)
    L.call("ttk_conv_fwd", p(d_a), p(a_bound), p(w_f), p(y), p(part), None, B, H, H, Cin, Cout, k, k, stride, pad)
    torch.cuda.synchronize()
    conv = lambda x, ww: F.conv2d(x, ww, stride=stride, padding=pad)
    judge("fwd", y.cpu().numpy(), nhwc(conv(a64, w64)), nhwc(conv(a32, w32)), nhwc(conv(fa, w64.abs()) + conv(a64.abs(), fw)))

    convT = lambda x, ww: F.conv_transpose2d(x, ww, stride=stride, padding=pad)
    g_in = torch.empty(B, H, H, Cin, device=dev)
    y0 = torch.zeros_like(y)
    L.call("ttk_conv_bwd_data", p(d_g), p(y0), p(d_bn), p(w_b), None, None, p(g_in), None, B, H, H, Cin, Cout, k, k, stride, pad)
    torch.cuda.synchronize()
    judge("dgrad", g_in.cpu().numpy(), nhwc(convT(g64, w64)), nhwc(convT(g32, w32)), nhwc(convT(fg, w64.abs()) + convT(g64.abs(), fw)))

    def wgrad(x, gg):  # dW[co][ci][kh][kw] = sum over pixels of gg (x) shifted x
        return torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), gg, stride=stride, padding=pad)
    dw = torch.zeros(Cout, Cin, k, k, device=dev)
    nb = L.conv_wgrad_partial_bytes(B, H, H, Cin, Cout, k, stride)
    scratch = torch.empty(max(nb // 4, 1), device=dev) if nb else None
    L.call("ttk_conv_bwd_weight", p(d_g), p(y0), p(d_bn), p(d_a), p(a_bound), p(dw), p(scratch), B, H, H, Cin, Cout, k, k, stride, pad)
    torch.cuda.synchronize()
    judge("wgrad", dw.cpu().numpy(), wgrad(a64, g64).numpy(), wgrad(a32, g32).numpy(), (wgrad(fa, g64.abs()) + wgrad(a64.abs(), fg)).numpy())
