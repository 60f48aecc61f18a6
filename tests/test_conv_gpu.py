"""GPU parity of the implicit-GEMM convolution entry points (ResNet18 variant) against float64 torch convolutions
on the host.  Inputs are channels-last activations, exactly as the backbone stores them.  The operand magnitude bounds
the fp16-split kernels scale by (include/ttk.h, row TTK_BN_AUX) are given with some slack, as the training step's are."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BN_SCALE, BN_BETA, BN_MEAN, BN_RSTD, BN_GA, BN_GB, BN_GMEAN, BN_AUX = range(8)
AUX_ACT_BOUND, AUX_DY_BOUND, AUX_GMAX = 0, 1, 2
F16 = __import__("os").environ.get("TTK_GEMM") != "bf16x3"  # the default fp16-pipe kernels (bounds, materialised dy, slice-wise weight gradient)


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


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# (B, H, Cin, Cout, k, stride): every conv shape class of ResNet18 at 129x129 input + ragged row counts
SHAPES = [(3, 33, 64, 64, 3, 1), (2, 33, 64, 128, 3, 2), (2, 33, 64, 128, 1, 2), (3, 17, 128, 128, 3, 1), (2, 17, 128, 256, 3, 2),
          (5, 9, 256, 256, 3, 1), (2, 9, 256, 512, 1, 2), (7, 5, 512, 512, 3, 1), (1, 9, 256, 512, 3, 2)]


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride", SHAPES)
def test_conv_fwd_bwd_data_bwd_weight(B, H, Cin, Cout, k, stride):
    import trackertraincode._hip as Hh
    L, p = Hh.lib(), Hh.ptr
    rng = np.random.default_rng(B * 1000 + H + Cin + Cout + k + stride)
    pad, W = k // 2, H
    Ho = (H + 2 * pad - k) // stride + 1
    a = np.maximum(rng.normal(0, 1, (B, H, W, Cin)), 0).astype(np.float32)  # a post-ReLU activation
    w = (rng.normal(0, 1, (Cout, Cin, k, k)) * np.sqrt(2.0 / (k * k * Cout))).astype(np.float32)
    dev = "cuda"
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_a, d_w = t(a), t(w)
    LOOSE = float(__import__("os").environ.get("LOOSE", "1.7"))
    a_bound = torch.tensor([LOOSE * float(a.max())], device=dev)
    w_f = torch.empty(3, k * k, Cout, Cin, dtype=torch.int16, device=dev)  # pre-split weight operands (6 bytes per weight)
    w_b = torch.empty(3, k * k, Cin, Cout, dtype=torch.int16, device=dev)
    L.call("ttk_conv_weight_repack", p(d_w), p(w_f), p(w_b), Cout, Cin, k, k)

    # ---- forward
    a64 = torch.from_numpy(a).double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = torch.from_numpy(w).double().requires_grad_(True)
    y64 = F.conv2d(a64, w64, stride=stride, padding=pad)
    y_ref = y64.detach().permute(0, 2, 3, 1).numpy()
    M = B * Ho * Ho
    rows = L.partial_rows_gemm(M)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    part = torch.full((rows, 2, Cout), float("nan"), device=dev)
    piv = rng.normal(0, 0.5, Cout).astype(np.float32)  # statistics pivot (include/ttk.h)
    d_piv = t(piv)
    L.call("ttk_conv_fwd", p(d_a), p(a_bound), p(w_f), p(y), p(part), p(d_piv), B, H, W, Cin, Cout, k, k, stride, pad)
    torch.cuda.synchronize()
    assert _rel(y.cpu().numpy(), y_ref) < 1.5e-6
    ps = part.cpu().numpy().astype(np.float64)
    assert np.isfinite(ps).all()
    flat = y_ref.reshape(-1, Cout) - piv.astype(np.float64)
    np.testing.assert_allclose(ps[:, 0].sum(0), flat.sum(0), rtol=0, atol=3e-5 * np.abs(flat).sum(0).max())
    np.testing.assert_allclose(ps[:, 1].sum(0), (flat ** 2).sum(0), rtol=3e-5)

    # ---- data gradient, raw and masked
    g = rng.normal(0, 1, (B, Ho, Ho, Cout)).astype(np.float32)
    bn = _bn_block(Cout, rng)
    yv = y.cpu().numpy()
    dy = (bn[BN_GA] * (g - bn[BN_GMEAN]) + bn[BN_GB] * (yv - bn[BN_MEAN])).astype(np.float32)
    bn[BN_AUX, AUX_DY_BOUND] = LOOSE * float(np.abs(dy).max())
    dy64 = torch.from_numpy(dy).double().permute(0, 3, 1, 2)
    ga_ref, gw_ref = torch.autograd.grad(y64, (a64, w64), dy64)
    ga_ref = ga_ref.permute(0, 2, 3, 1).numpy()
    d_g, d_bn = t(g), t(bn)
    g_in = torch.empty(B, H, W, Cin, device=dev)
    L.call("ttk_conv_bwd_data", p(d_g), p(y), p(d_bn), p(w_b), None, None, p(g_in), None, B, H, W, Cin, Cout, k, k, stride, pad)
    torch.cuda.synchronize()
    assert _rel(g_in.cpu().numpy(), ga_ref) < 1.5e-6
    if Cin % 64 == 0:
        mask_y = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float32)
        mbn = _bn_block(Cin, rng)
        pre = mbn[BN_SCALE] * (mask_y - mbn[BN_MEAN]) + mbn[BN_BETA]
        safe = np.abs(pre) > 1e-4
        ref = ga_ref * (pre > 0)
        d_my, d_mbn = t(mask_y), t(mbn)
        rows_in = L.partial_rows_gemm(B * H * W)
        part2 = torch.full((rows_in, 2, Cin), float("nan"), device=dev)
        L.call("ttk_conv_bwd_data", p(d_g), p(y), p(d_bn), p(w_b), p(d_my), p(d_mbn), p(g_in), p(part2), B, H, W, Cin, Cout, k, k,
               stride, pad)
        torch.cuda.synchronize()
        out = g_in.cpu().numpy()
        assert _rel(out * safe, ref * safe) < 1.5e-6
        if F16:
            assert float(d_mbn[BN_AUX, AUX_GMAX]) == float(np.abs(out).max())  # the bound of the next layer's gradient operand
        ps = part2.cpu().numpy().astype(np.float64)
        o64 = out.astype(np.float64).reshape(-1, Cin)
        np.testing.assert_allclose(ps[:, 0].sum(0), o64.sum(0), rtol=0, atol=3e-5 * np.abs(o64).sum(0).max())
        yc = (mask_y.astype(np.float64) - mbn[BN_MEAN]).reshape(-1, Cin)
        np.testing.assert_allclose(ps[:, 1].sum(0), (o64 * yc).sum(0), rtol=0, atol=3e-5 * np.abs(o64 * yc).sum(0).max())

    # ---- weight gradient (accumulates onto a zeroed buffer; torch layout [Cout][Cin][k][k]), atomic form
    dw = torch.zeros(Cout, Cin, k, k, device=dev)
    L.call("ttk_conv_bwd_weight", p(d_g), p(y), p(d_bn), p(d_a), p(a_bound), p(dw), None, B, H, W, Cin, Cout, k, k, stride, pad)
    torch.cuda.synchronize()
    assert _rel(dw.cpu().numpy(), gw_ref.numpy()) < 1.5e-6
    if not F16:
        return

    # ---- the same two gradients from a materialised dy (ttk_bn_bwd_apply; y == NULL)
    d_dy = torch.empty_like(d_g)
    L.call("ttk_bn_bwd_apply", p(d_g), p(y), p(d_bn), p(d_dy), M, Cout)
    torch.cuda.synchronize()
    # dy arrives as two fp16 planes [M][Cout] (h, then l) of dy * 2^s, 2^s * bound in [2^14, 2^15): h + l reproduces it to fp32 rounding
    S = 2.0 ** (14 - int(np.floor(np.log2(float(bn[BN_AUX, AUX_DY_BOUND])))))
    planes = d_dy.view(torch.float16).reshape(2, M, Cout).double().cpu().numpy()
    np.testing.assert_allclose((planes[0] + planes[1]) / S, dy.reshape(M, Cout), rtol=2e-6, atol=1e-6 * float(np.abs(dy).max()))
    L.call("ttk_conv_bwd_data", p(d_dy), None, p(d_bn), p(w_b), None, None, p(g_in), None, B, H, W, Cin, Cout, k, k, stride, pad)
    dw2 = torch.zeros(Cout, Cin, k, k, device=dev)
    nb = L.conv_wgrad_partial_bytes(B, H, W, Cin, Cout, k, stride)
    assert (nb > 0) == (k == 3)  # 1x1 kernels keep the (coalesced) atomics
    scratch = torch.full((nb // 4,), float("nan"), device=dev) if nb else None  # slice-wise, atomic-free form
    L.call("ttk_conv_bwd_weight", p(d_dy), None, p(d_bn), p(d_a), p(a_bound), p(dw2), p(scratch), B, H, W, Cin, Cout, k, k, stride, pad)
    torch.cuda.synchronize()
    assert _rel(g_in.cpu().numpy(), ga_ref) < 1.5e-6
    assert _rel(dw2.cpu().numpy(), gw_ref.numpy()) < 1.5e-6
    dw3 = torch.zeros_like(dw2)  # the slice-wise form is bitwise reproducible
    L.call("ttk_conv_bwd_weight", p(d_dy), None, p(d_bn), p(d_a), p(a_bound), p(dw3), p(scratch), B, H, W, Cin, Cout, k, k, stride, pad)
    assert k == 1 or torch.equal(dw2, dw3)

