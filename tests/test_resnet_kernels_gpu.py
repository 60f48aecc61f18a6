"""GPU parity of the non-GEMM ResNet18 kernels (csrc/resnet.hip) against float64 torch ops on the host."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BN_SCALE, BN_BETA, BN_MEAN, BN_RSTD, BN_GA, BN_GB, BN_GMEAN = range(7)


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


def _pre(bn, y):
    return bn[BN_SCALE] * (y - bn[BN_MEAN]) + bn[BN_BETA]


def _sums_close(part, v, yc):
    ps = part.cpu().numpy().astype(np.float64)
    assert np.isfinite(ps).all()
    v, yc = v.reshape(-1, v.shape[-1]).astype(np.float64), yc.reshape(-1, yc.shape[-1]).astype(np.float64)
    np.testing.assert_allclose(ps[:, 0].sum(0), v.sum(0), rtol=0, atol=3e-5 * np.abs(v).sum(0).max() + 1e-12)
    np.testing.assert_allclose(ps[:, 1].sum(0), (v * yc).sum(0), rtol=0, atol=3e-5 * np.abs(v * yc).sum(0).max() + 1e-12)


@pytest.mark.parametrize("B,H", [(3, 129), (2, 64), (1, 33)])
def test_stem7(B, H):
    import trackertraincode._hip as Hh
    L, p = Hh.lib(), Hh.ptr
    rng = np.random.default_rng(B + H)
    x = rng.uniform(-0.5, 0.5, (B, 1, H, H)).astype(np.float32)
    w = (rng.normal(0, 1, (64, 1, 7, 7)) * 0.2).astype(np.float32)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x64, w64 = torch.from_numpy(x).double(), torch.from_numpy(w).double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, stride=2, padding=3)
    Ho = y64.shape[-1]
    y_ref = y64.detach().permute(0, 2, 3, 1).numpy()
    rows = L.partial_rows_elementwise(B * Ho * Ho * 16)
    d_x, d_w = t(x), t(w)
    y = torch.empty(B, Ho, Ho, 64, device=dev)
    part = torch.full((rows, 2, 64), float("nan"), device=dev)
    piv = rng.normal(0, 0.1, 64).astype(np.float32)  # statistics pivot (include/ttk.h)
    d_piv = t(piv)
    L.call("ttk_stem7_fwd", p(d_x), p(d_w), p(y), p(part), p(d_piv), B, H, H)
    torch.cuda.synchronize()
    assert _rel(y.cpu().numpy(), y_ref) < 1e-6
    ps = part.cpu().numpy().astype(np.float64)
    flat = y_ref.reshape(-1, 64) - piv.astype(np.float64)
    np.testing.assert_allclose(ps[:, 0].sum(0), flat.sum(0), rtol=0, atol=3e-5 * np.abs(flat).sum(0).max())
    np.testing.assert_allclose(ps[:, 1].sum(0), (flat ** 2).sum(0), rtol=3e-5)
    g = rng.normal(0, 1, (B, Ho, Ho, 64)).astype(np.float32)
    bn = _bn_block(64, rng)
    dy = (bn[BN_GA] * (g - bn[BN_GMEAN]) + bn[BN_GB] * (y.cpu().numpy() - bn[BN_MEAN])).astype(np.float32)
    (gw_ref,) = torch.autograd.grad(y64, w64, torch.from_numpy(dy).double().permute(0, 3, 1, 2))
    d_g, d_bn = t(g), t(bn)
    dw = torch.full((64, 1, 7, 7), float("nan"), device=dev)
    L.call("ttk_stem7_bwd_weight", p(d_g), p(y), p(d_bn), p(d_x), p(dw), None, B, H, H)
    torch.cuda.synchronize()
    assert _rel(dw.cpu().numpy(), gw_ref.numpy()) < 2e-6
    # deterministic form: workgroup partials folded in a fixed order - bitwise equal from run to run
    nb = L.cdll.ttk_stem7_wgrad_partial_bytes(B, H, H)
    scratch = torch.full((nb // 4,), float("nan"), device=dev)
    outs = []
    for _ in range(2):
        dwp = torch.full((64, 1, 7, 7), float("nan"), device=dev)
        L.call("ttk_stem7_bwd_weight", p(d_g), p(y), p(d_bn), p(d_x), p(dwp), p(scratch), B, H, H)
        torch.cuda.synchronize()
        outs.append(dwp)
    assert torch.equal(outs[0], outs[1]) and _rel(outs[0].cpu().numpy(), gw_ref.numpy()) < 2e-6


@pytest.mark.parametrize("B,H,C", [(3, 65, 64), (2, 17, 32), (1, 8, 128)])
def test_maxpool(B, H, C):
    import trackertraincode._hip as Hh
    L, p = Hh.lib(), Hh.ptr
    rng = np.random.default_rng(B + H + C)
    y = rng.normal(0, 1, (B, H, H, C)).astype(np.float32)
    bn = _bn_block(C, rng)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    s64 = torch.from_numpy(_pre(bn, y).astype(np.float32)).double().permute(0, 3, 1, 2).requires_grad_(True)
    a64 = F.max_pool2d(torch.relu(s64), 3, 2, 1)
    Ho = a64.shape[-1]
    d_y, d_bn = t(y), t(bn)
    a = torch.empty(B, Ho, Ho, C, device=dev)
    idx = torch.empty(B, Ho, Ho, C, dtype=torch.uint8, device=dev)
    L.call("ttk_maxpool3x3s2_fwd", p(d_y), p(d_bn), p(a), p(idx), B, H, H, C)
    torch.cuda.synchronize()
    np.testing.assert_allclose(a.cpu().numpy(), a64.detach().permute(0, 2, 3, 1).numpy(), rtol=1e-6, atol=1e-6)
    assert float(d_bn[7, 0]) == float(a.max())  # TTK_AUX_ACT_BOUND: what the first residual block's convolutions scale by
    ga, gb = rng.normal(0, 1, (B, Ho, Ho, C)).astype(np.float32), rng.normal(0, 1, (B, Ho, Ho, C)).astype(np.float32)
    (gs_ref,) = torch.autograd.grad(a64, s64, torch.from_numpy(ga + gb).double().permute(0, 3, 1, 2))
    gs_ref = gs_ref.permute(0, 2, 3, 1).numpy()
    rows = L.partial_rows_elementwise(B * H * H * (C // 4))
    d_ga, d_gb = t(ga), t(gb)
    g = torch.empty(B, H, H, C, device=dev)
    part = torch.full((rows, 2, C), float("nan"), device=dev)
    L.call("ttk_maxpool3x3s2_bwd", p(d_ga), p(d_gb), p(idx), p(d_y), p(d_bn), p(g), p(part), B, H, H, C)
    torch.cuda.synchronize()
    np.testing.assert_allclose(g.cpu().numpy(), gs_ref, rtol=1e-5, atol=1e-6)
    _sums_close(part, g.cpu().numpy(), y - bn[BN_MEAN])


@pytest.mark.parametrize("rows,C", [(1000, 64), (77, 512), (4096, 128)])
def test_bn_add_act_and_residual_bwd(rows, C):
    import trackertraincode._hip as Hh
    L, p = Hh.lib(), Hh.ptr
    rng = np.random.default_rng(rows + C)
    y, yd, act = (rng.normal(0, 1, (rows, C)).astype(np.float32) for _ in range(3))
    act = np.maximum(act, 0)
    bn, bnd = _bn_block(C, rng), _bn_block(C, rng)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_y, d_yd, d_act, d_bn, d_bnd = t(y), t(yd), t(act), t(bn), t(bnd)
    out = torch.empty(rows, C, device=dev)
    for res, res_bn, ref in ((None, None, _pre(bn, y)), (d_act, None, _pre(bn, y) + act), (d_yd, d_bnd, _pre(bn, y) + _pre(bnd, yd))):
        d_bn[7, 0] = 0.0
        L.call("ttk_bn_add_act", p(d_y), p(d_bn), p(res), p(res_bn), p(out), None, 1, rows, C)  # measure: eval mode
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), np.maximum(ref, 0), rtol=1e-5, atol=1e-6)
        assert float(d_bn[7, 0]) == float(out.max())  # TTK_AUX_ACT_BOUND raised to the maximum of what was written
        d_bn[7, 0] = 2.5  # training: the bound of relu(bn(y)) from the statistics + the bound of the residual
        rb = torch.tensor([1.25], device=dev)
        L.call("ttk_bn_add_act", p(d_y), p(d_bn), p(res), p(res_bn), p(out), p(rb) if res is not None else None, 0, rows, C)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), np.maximum(ref, 0), rtol=1e-5, atol=1e-6)
        assert float(d_bn[7, 0]) == (3.75 if res is not None else 2.5)
    ga, gb = rng.normal(0, 1, (rows, C)).astype(np.float32), rng.normal(0, 1, (rows, C)).astype(np.float32)
    d_ga, d_gb = t(ga), t(gb)
    prow = L.partial_rows_elementwise(rows * (C // 4))
    gs = torch.empty(rows, C, device=dev)
    part, partd = torch.full((prow, 2, C), float("nan"), device=dev), torch.full((prow, 2, C), float("nan"), device=dev)
    L.call("ttk_residual_bwd", p(d_ga), p(d_gb), p(d_act), p(d_y), p(d_bn), p(d_yd), p(d_bnd), p(gs), p(part), p(partd), rows, C)
    torch.cuda.synchronize()
    ref = (ga + gb) * (act > 0)
    np.testing.assert_allclose(gs.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    _sums_close(part, ref, y - bn[BN_MEAN])
    _sums_close(partd, ref, yd - bnd[BN_MEAN])
    assert float(d_bn[7, 2]) == float(d_bnd[7, 2]) == float(gs.abs().max())  # TTK_AUX_GMAX of both BatchNorms
    L.call("ttk_residual_bwd", p(d_ga), None, p(d_act), p(d_y), p(d_bn), None, None, p(gs), p(part), None, rows, C)
    torch.cuda.synchronize()
    np.testing.assert_allclose(gs.cpu().numpy(), ga * (act > 0), rtol=1e-6, atol=1e-6)
