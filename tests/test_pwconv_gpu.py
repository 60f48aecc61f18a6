"""GPU parity of the pointwise-conv GEMM entry points (C-ABI) against float64 numpy.

The compute-bound shapes run on the 16-bit matrix pipe with split operands (csrc/pwconv_f16.hip: two fp16 pieces and
three products, scaled by the magnitude bounds of row TTK_BN_AUX; TTK_GEMM=bf16x3: csrc/pwconv_split.hip); the criterion
is that they are as close to the exact product as a chain of fp32 multiply-adds over the same operands is (one fp32
accumulator per output, k by k) - i.e. no precision was given up for the speed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BN_SCALE, BN_BETA, BN_MEAN, BN_RSTD, BN_GA, BN_GB, BN_GMEAN, BN_AUX = range(8)
AUX_ACT_BOUND, AUX_DY_BOUND, AUX_GMAX = range(3)


def _bn_block(C, rng):
    bn = np.zeros((8, C), np.float32)
    bn[BN_SCALE] = rng.uniform(0.5, 1.5, C)
    bn[BN_BETA] = rng.normal(0, 0.2, C)
    bn[BN_MEAN] = rng.normal(0, 0.3, C)
    bn[BN_RSTD] = rng.uniform(0.5, 2.0, C)
    bn[BN_GA] = rng.uniform(0.5, 1.5, C)
    bn[BN_GB] = rng.normal(0, 0.2, C)
    bn[BN_GMEAN] = rng.normal(0, 0.05, C)
    return bn


def _chain32(a, b_t):
    """a[M,K] @ b_t[K,N] accumulated k by k in float32 (what a chain of fp32 fused multiply-adds gives, up to the
    rounding of the product): the accuracy class of any GEMM that keeps ONE fp32 accumulator per output."""
    acc = np.zeros((a.shape[0], b_t.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc += a[:, k:k + 1] * b_t[k:k + 1, :]
    return acc


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# (M, Cin, Cout): split-kernel tiles 128x256 and 256x128, ragged M, the fp32-MFMA shapes, wide K
SHAPES = [(648, 512, 512), (1000, 128, 256), (4100, 256, 256), (300, 1024, 1024), (777, 256, 128), (2049, 512, 128),
          (648, 64, 128), (1234, 32, 64), (5000, 128, 128), (128, 512, 1024),
          # row-block kernels (csrc/pwconv_r.hip) at tile heights their cost model picks by itself: 162 of 192 rows (two column tiles:
          # one full round of 256 CUs), 200 of 256 rows, 193-row blocks with a ragged last one
          (20736, 512, 512), (12800, 256, 1024), (49601, 128, 256),
          # the streaming kernel of the narrow layers (csrc/pwconv_y.hip: 32 -> 64, 64 -> 128, 128 -> 128 forward; data gradient of 128 -> 256): more pixel
          # groups than resident waves, a ragged last group, fewer groups than waves
          (140001, 32, 64), (99990, 64, 128), (70001, 128, 128), (33, 64, 128), (41111, 128, 256)]


# LOOSE: how far the operand bounds of row TTK_BN_AUX lie above the true maxima (the step's own bounds are 1-100x loose)
@pytest.mark.parametrize("loose", [1.0, 300.0])
@pytest.mark.parametrize("M,Cin,Cout", SHAPES)
def test_pwconv_fwd_bwd_data_bwd_weight(M, Cin, Cout, loose):
    import trackertraincode._hip as H
    L, p = H.lib(), H.ptr
    rng = np.random.default_rng(M + Cin + Cout)
    ydw = rng.normal(0, 1, (M, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin)) * np.sqrt(2.0 / Cout)).astype(np.float32)
    bn_dw, bn_pw = _bn_block(Cin, rng), _bn_block(Cout, rng)
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)
    rows, rows_b = L.partial_rows_gemm(M, Cin, Cout), L.partial_rows_gemm(M, Cout, Cin, True)  # forward: K = Cin, N = Cout; data gradient: K = Cout, N = Cin

    # ---- forward: y = relu(scale*(ydw-mean)+beta) @ w^T ; partial sums of y and y^2 per column
    a32 = np.maximum(bn_dw[BN_SCALE] * (ydw - bn_dw[BN_MEAN]) + bn_dw[BN_BETA], 0).astype(np.float32)
    bn_dw[BN_AUX, AUX_ACT_BOUND] = np.abs(a32).max() * loose
    y64 = a32.astype(np.float64) @ w.astype(np.float64).T
    y32 = _chain32(a32, np.ascontiguousarray(w.T))
    d_ydw, d_w, d_bn = H.to_blocks(t(ydw)), t(w), t(bn_dw)  # activations travel as channel blocks (include/ttk.h)
    y = torch.empty(M, Cout, device=dev)
    part = torch.full((rows, 2, Cout), float("nan"), device=dev)
    wq = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device=dev)  # scratch for the split weight operand
    piv = rng.normal(0, 0.5, Cout).astype(np.float32)  # statistics pivot (include/ttk.h): the partial sums are those of y - pivot
    d_piv = t(piv)
    L.call("ttk_pwconv1x1_fwd", p(d_ydw), p(d_bn), p(d_w), p(y), p(part), p(d_piv), M, Cin, Cout, p(wq), 0)
    torch.cuda.synchronize()
    e_hip, e_f32 = _rel(H.from_blocks(y).cpu().numpy(), y64), _rel(y32, y64)
    print(f"fwd   M={M} K={Cin} N={Cout}: hip {e_hip:.2e}  fp32 chain {e_f32:.2e}")
    assert e_hip <= 1.5 * e_f32 + 1e-7, (e_hip, e_f32)
    ps = part.cpu().numpy().astype(np.float64)
    assert np.isfinite(ps).all()
    ys = y64 - piv.astype(np.float64)
    np.testing.assert_allclose(ps[:, 0].sum(0), ys.sum(0), rtol=0, atol=2e-5 * np.abs(ys).sum(0).max())
    np.testing.assert_allclose(ps[:, 1].sum(0), (ys ** 2).sum(0), rtol=2e-5)

    # ---- data gradient: g_dw = (dy @ w) * [bn_dw(ydw) > 0], dy = ga*(g-gmean) + gb*(y-mean_pw)
    g = rng.normal(0, 1, (M, Cout)).astype(np.float32)
    yv = H.from_blocks(y).cpu().numpy()
    dy32 = (bn_pw[BN_GA] * (g - bn_pw[BN_GMEAN]) + bn_pw[BN_GB] * (yv - bn_pw[BN_MEAN])).astype(np.float32)
    bn_pw[BN_AUX, AUX_DY_BOUND] = np.abs(dy32).max() * loose
    pre = bn_dw[BN_SCALE] * (ydw - bn_dw[BN_MEAN]) + bn_dw[BN_BETA]
    mask = pre > 0
    safe = np.abs(pre) > 1e-4  # entries whose mask could flip with rounding are left out of the comparison
    gd64 = (dy32.astype(np.float64) @ w.astype(np.float64)) * mask
    gd32 = _chain32(dy32, w) * mask
    wt = torch.from_numpy(np.ascontiguousarray(w.T)).to(dev)
    g_dw = torch.empty(M, Cin, device=dev)
    part2 = torch.full((rows_b, 2, Cin), float("nan"), device=dev)
    d_g, d_bnpw = H.to_blocks(t(g)), t(bn_pw)  # named: a temporary would be recycled by the allocator before the kernel runs
    L.call("ttk_pwconv1x1_bwd_data", p(d_g), p(y), p(d_bnpw), p(wt), p(d_ydw), p(d_bn), p(g_dw), p(part2), M, Cin, Cout, p(wq), 0)
    torch.cuda.synchronize()
    out = H.from_blocks(g_dw).cpu().numpy()
    e_hip, e_f32 = _rel(out * safe, gd64 * safe), _rel(gd32 * safe, gd64 * safe)
    print(f"dgrad M={M} K={Cout} N={Cin}: hip {e_hip:.2e}  fp32 chain {e_f32:.2e}")
    assert e_hip <= 1.5 * e_f32 + 1e-7, (e_hip, e_f32)
    ps = part2.cpu().numpy().astype(np.float64)
    assert np.isfinite(ps).all()
    o64 = out.astype(np.float64)
    np.testing.assert_allclose(ps[:, 0].sum(0), o64.sum(0), rtol=0, atol=2e-5 * np.abs(o64).sum(0).max())
    s2 = (o64 * (ydw.astype(np.float64) - bn_dw[BN_MEAN])).sum(0)
    np.testing.assert_allclose(ps[:, 1].sum(0), s2, rtol=0, atol=2e-5 * np.abs(o64 * (ydw - bn_dw[BN_MEAN])).sum(0).max())

    # ---- weight gradient: dW[co][ci] = sum_m dy[m][co] * a[m][ci] (accumulated onto a zeroed buffer)
    dw64 = dy32.astype(np.float64).T @ a32.astype(np.float64)
    dw32 = _chain32(np.ascontiguousarray(dy32.T), a32)
    dW = torch.zeros(Cout, Cin, device=dev)
    L.call("ttk_pwconv1x1_bwd_weight", p(d_g), p(y), p(d_bnpw), p(d_ydw), p(d_bn), p(dW), None, M, Cin, Cout, 0)
    torch.cuda.synchronize()
    e_hip, e_f32 = _rel(dW.cpu().numpy(), dw64), _rel(dw32, dw64)
    print(f"wgrad M={M} Cout={Cout} Cin={Cin}: hip {e_hip:.2e}  fp32 chain {e_f32:.2e}")
    assert e_hip <= 1.5 * e_f32 + 1e-7, (e_hip, e_f32)
    # deterministic form: slices of M stored to scratch and folded in a fixed order - bitwise reproducible
    nbytes = L.pwconv_wgrad_partial_bytes(M, Cin, Cout)
    if nbytes:
        runs = []
        for _ in range(2):
            scratch = torch.full((nbytes // 4,), float("nan"), device=dev)
            dWd = torch.zeros(Cout, Cin, device=dev)
            L.call("ttk_pwconv1x1_bwd_weight", p(d_g), p(y), p(d_bnpw), p(d_ydw), p(d_bn), p(dWd), p(scratch), M, Cin, Cout, 0)
            torch.cuda.synchronize()
            runs.append(dWd)
        assert torch.equal(runs[0], runs[1])
        assert _rel(runs[0].cpu().numpy(), dw64) <= 1.5 * e_f32 + 1e-7


@pytest.mark.parametrize("tile_rows", [128, 192, 256])
@pytest.mark.parametrize("Cin,Cout", [(256, 512), (512, 256)])
def test_row_block_gemm_every_tile_height(Cin, Cout, tile_rows):
    """The row-block GEMMs (pw16r_k / pw16m_k) at each of their three tile heights: forward and data gradient against float64, partial
    sums, ragged last row block.  The cost model picks the height from (M, K, Nout); ttk_pwconv_tile_rows (ABI 19) reports it, and the
    test searches an M for which BOTH directions run the wanted height (the product library has no switch that forces one)."""
    import trackertraincode._hip as H
    L = H.lib()
    found = None
    for M in list(range(1000, 20000, 37)) + list(range(20000, 120000, 997)):
        if L.cdll.ttk_pwconv_tile_rows(M, Cin, Cout, 0) == tile_rows and L.cdll.ttk_pwconv_tile_rows(M, Cout, Cin, 1) == tile_rows:
            found = M
            break
    if found is None:  # the two directions need not agree on one M: take one M per direction
        ms = []
        for dgrad, (k, n) in enumerate(((Cin, Cout), (Cout, Cin))):
            ms.append(next((M for M in list(range(1000, 20000, 37)) + list(range(20000, 120000, 997)) if L.cdll.ttk_pwconv_tile_rows(M, k, n, dgrad) == tile_rows), None))
        assert all(m is not None for m in ms), f"no M in the searched range runs {tile_rows}-row tiles for {Cin}->{Cout}: {ms}"
        for M in sorted(set(ms)):
            test_pwconv_fwd_bwd_data_bwd_weight(M, Cin, Cout, 1.0)
    else:
        test_pwconv_fwd_bwd_data_bwd_weight(found, Cin, Cout, 1.0)


def test_prepared_weights_match_per_call_split():
    """ttk_pwconv_prepare_weights (all layers, one launch) feeds the same kernels the same operand bits as the
    per-call split/transposition: outputs are bit-identical."""
    import trackertraincode._hip as H
    L, p = H.lib(), H.ptr
    dev = "cuda"
    shapes = [(648, 512, 512), (1234, 32, 64), (777, 256, 128), (300, 1024, 1024), (5000, 128, 128), (900, 64, 128)]
    rng = np.random.default_rng(5)
    ws = [torch.from_numpy((rng.normal(0, 1, (co, ci, 1, 1)) * np.sqrt(2.0 / co)).astype(np.float32)).to(dev) for _, ci, co in shapes]
    prep = [torch.empty(L.pwconv_prepared_bytes(ci, co), dtype=torch.uint8, device=dev) for _, ci, co in shapes]
    L.pwconv_prepare_weights(ws, prep)
    for (M, Cin, Cout), w, q in zip(shapes, ws, prep):
        ydw = torch.from_numpy(rng.normal(0, 1, (M, Cin)).astype(np.float32)).to(dev)
        g = torch.from_numpy(rng.normal(0, 1, (M, Cout)).astype(np.float32)).to(dev)
        bn_dw, bn_pw = _bn_block(Cin, rng), _bn_block(Cout, rng)
        bn_dw[BN_AUX, AUX_ACT_BOUND], bn_pw[BN_AUX, AUX_DY_BOUND] = 12.0, 40.0  # generous for N(0,1) data with these constants
        bn_dw, bn_pw = torch.from_numpy(bn_dw).to(dev), torch.from_numpy(bn_pw).to(dev)
        rows, rows_b = L.partial_rows_gemm(M, Cin, Cout), L.partial_rows_gemm(M, Cout, Cin, True)
        wq = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
        wt = w.reshape(Cout, Cin).t().contiguous()
        out = []
        for prepared in (False, True):
            y, part = torch.empty(M, Cout, device=dev), torch.zeros(rows, 2, Cout, device=dev)
            L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), None if prepared else p(w), p(y), p(part), None, M, Cin, Cout, p(q if prepared else wq), 0)
            gd, part2 = torch.empty(M, Cin, device=dev), torch.zeros(rows_b, 2, Cin, device=dev)
            L.call("ttk_pwconv1x1_bwd_data", p(g), p(y), p(bn_pw), None if prepared else p(wt), p(ydw), p(bn_dw), p(gd), p(part2), M, Cin, Cout,
                   p(q if prepared else wq), 0)
            torch.cuda.synchronize()
            out.append((y, part, gd, part2))
        for a, b in zip(*out):
            assert torch.equal(a, b), (M, Cin, Cout)
    with pytest.raises(RuntimeError, match="null pointer"):
        L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), None, p(y), None, None, M, Cin, Cout, None, 0)


@pytest.mark.parametrize("M,Cin,Cout", [(64 * 300 + 17, 32, 64), (64 * 1100 + 63, 32, 64), (64 * 200 + 1, 64, 128), (64 * 700 + 40, 64, 128), (50, 64, 128),
                                        (32 * 500 + 9, 128, 128), (32 * 1300 + 31, 128, 128), (20, 128, 128)])
def test_fused_bwd_matches_fp64_and_the_two_kernels(M, Cin, Cout):
    """ttk_pwconv1x1_bwd_fused (first two pointwise layers: weight + data gradient from one read of the operands) against
    float64 numpy, against the two kernels it replaces, and - with the scratch buffer - bitwise reproducible."""
    import trackertraincode._hip as Hh
    L, p = Hh.lib(), Hh.ptr
    rng = np.random.default_rng(M + Cin)
    g = (rng.normal(0, 1, (M, Cout)) * 1e-2).astype(np.float32)
    y = rng.normal(0, 1, (M, Cout)).astype(np.float32)
    ydw = rng.normal(0, 1, (M, Cin)).astype(np.float32)
    w = (rng.normal(0, 1, (Cout, Cin)) * np.sqrt(2.0 / Cout)).astype(np.float32)
    bn_pw, bn_dw = _bn_block(Cout, rng), _bn_block(Cin, rng)
    dy = bn_pw[BN_GA].astype(np.float64) * (g - bn_pw[BN_GMEAN]) + bn_pw[BN_GB].astype(np.float64) * (y - bn_pw[BN_MEAN])
    yc = ydw.astype(np.float64) - bn_dw[BN_MEAN]
    pre = bn_dw[BN_SCALE] * yc + bn_dw[BN_BETA]
    gdw_ref = (dy @ w.astype(np.float64)) * (pre > 0)
    dw_ref = dy.T @ np.maximum(pre, 0)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_g, d_y, d_ydw, d_w, d_bnpw, d_bndw = Hh.to_blocks(t(g)), Hh.to_blocks(t(y)), Hh.to_blocks(t(ydw)), t(w), t(bn_pw), t(bn_dw)  # activations: channel blocks
    prep = None
    if Cin >= 64:  # the fp16-pipe forms: prepared weight block (planes / |w| maximum) and the operand bounds of row TTK_BN_AUX (1.7x loose, as the step's are)
        prep = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
        L.pwconv_prepare_weights([d_w.view(Cout, Cin, 1, 1)], [prep])
        d_bnpw[BN_AUX, AUX_DY_BOUND] = 1.7 * float(np.abs(dy).max())
        d_bndw[BN_AUX, AUX_ACT_BOUND] = 1.7 * float(np.maximum(pre, 0).max())
    rows = L.cdll.ttk_pwconv1x1_bwd_fused_rows(M, Cin, Cout)
    assert rows > 0
    g_dw = torch.full((M, Cin), float("nan"), device=dev)
    part = torch.full((rows, 2, Cin), float("nan"), device=dev)
    dw = torch.zeros(Cout, Cin, device=dev)
    L.call("ttk_pwconv1x1_bwd_fused", p(d_g), p(d_y), p(d_bnpw), p(d_w), p(prep), p(d_ydw), p(d_bndw), p(g_dw), p(dw), None, p(part), M, Cin, Cout)
    torch.cuda.synchronize()
    safe = np.abs(pre) > 1e-4  # a pre-activation within rounding of zero may fall on either side of the ReLU
    out = Hh.from_blocks(g_dw).cpu().numpy()
    assert np.isfinite(out).all()
    tol = 2e-6 if Cin < 64 else 3e-6  # fp32 MFMA (exact products) | fp16 split (an fp32 fma chain's accuracy)
    assert _rel(out * safe, gdw_ref * safe) < tol
    assert _rel(dw.cpu().numpy(), dw_ref) < tol
    ps = part.cpu().numpy().astype(np.float64)
    o64 = out.astype(np.float64)
    np.testing.assert_allclose(ps[:, 0].sum(0), o64.sum(0), rtol=0, atol=3e-5 * np.abs(o64).sum(0).max())
    np.testing.assert_allclose(ps[:, 1].sum(0), (o64 * yc).sum(0), rtol=0, atol=3e-5 * np.abs(o64 * yc).sum(0).max())
    # the two kernels it replaces compute the same fp32 arithmetic (exact products, fp32 accumulation, another summation order)
    g_dw2 = torch.empty(M, Cin, device=dev)
    part2 = torch.empty(L.partial_rows_gemm(M, Cout, Cin, True), 2, Cin, device=dev)
    dw2 = torch.zeros(Cout, Cin, device=dev)
    wt = d_w.t().contiguous()
    wq2 = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device=dev)  # (named: see above)
    L.call("ttk_pwconv1x1_bwd_data", p(d_g), p(d_y), p(d_bnpw), p(wt), p(d_ydw), p(d_bndw), p(g_dw2), p(part2), M, Cin, Cout, p(wq2), 0)
    L.call("ttk_pwconv1x1_bwd_weight", p(d_g), p(d_y), p(d_bnpw), p(d_ydw), p(d_bndw), p(dw2), None, M, Cin, Cout, 0)
    torch.cuda.synchronize()
    assert _rel(out * safe, Hh.from_blocks(g_dw2).cpu().numpy() * safe) < tol and _rel(dw.cpu().numpy(), dw2.cpu().numpy()) < tol
    # deterministic form
    nb = L.cdll.ttk_pwconv1x1_bwd_fused_partial_bytes(M, Cin, Cout)
    scratch = torch.full((nb // 4,), float("nan"), device=dev)
    res = []
    for _ in range(2):
        dwp = torch.zeros(Cout, Cin, device=dev)
        L.call("ttk_pwconv1x1_bwd_fused", p(d_g), p(d_y), p(d_bnpw), p(d_w), p(prep), p(d_ydw), p(d_bndw), p(g_dw), p(dwp), p(scratch), p(part), M, Cin, Cout)
        torch.cuda.synchronize()
        res.append((dwp.clone(), g_dw.clone(), part.clone()))
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    assert _rel(res[0][0].cpu().numpy(), dw_ref) < tol
