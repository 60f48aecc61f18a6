"""Depthwise 3x3 forward / data-gradient entry points (C-ABI) against a float64 torch reference over the tiling paths of
csrc/dwconv_tiled.hip: row bands, several images per tile (small images, ragged last tile), column tiles (wide images,
ragged last column tile), stride 2 with odd and even sizes, residual inputs, materialised or recomputed block input."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SCALE, BETA, MEAN, RSTD, GA, GB, GMEAN = range(7)


def _bn(C, g):
    bn = torch.zeros(8, C, dtype=torch.float64)
    bn[SCALE] = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    bn[BETA] = torch.randn(C, generator=g, dtype=torch.float64) * 0.2
    bn[MEAN] = torch.randn(C, generator=g, dtype=torch.float64) * 0.3
    bn[RSTD] = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    bn[GA] = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    bn[GB] = torch.randn(C, generator=g, dtype=torch.float64) * 0.2
    bn[GMEAN] = torch.randn(C, generator=g, dtype=torch.float64) * 0.05
    return bn


def _nchw(t):  # [B,H,W,C] -> [B,C,H,W]
    return t.permute(0, 3, 1, 2)


# (B, H, W, C, stride, skip)
SHAPES = [(4, 9, 9, 64, 1, True),     # 3 images per tile, ragged last tile
          (5, 5, 5, 32, 1, False),    # 5x5: all images in one tile
          (9, 5, 5, 64, 1, True),     # 7 + 2 images
          (2, 50, 50, 32, 1, False),  # column tiles, last one narrower
          (2, 40, 70, 32, 1, True),   # non-square, bands and column tiles
          (2, 65, 65, 32, 1, False),  # the network's first layer
          (3, 33, 33, 64, 1, True),   # row bands only
          (2, 33, 33, 64, 2, False),  # stride 2, odd size
          (3, 10, 12, 32, 2, False),  # stride 2, even sizes
          (6, 9, 9, 128, 2, False),   # stride 2, several images per tile
          (1, 17, 17, 256, 1, True),
          # more bands than resident workgroups: a workgroup of the forward walks CONSECUTIVE bands of an image and carries their
          # shared input rows over in LDS (round 4), across image boundaries too; a_out rows are stored by whoever stages them
          (40, 65, 65, 32, 1, True),
          (40, 65, 65, 32, 2, False),
          (100, 33, 33, 64, 1, True),
          (90, 33, 33, 64, 2, False),
          # round 6: the BACKWARD walks consecutive full-width bands too (stride 1, staged width <= 80: a ring of LDS rows keeps the dy rows
          # two bands share) - the lean form (no residual operands) across image boundaries, and images too wide for it (column tiles stay)
          (40, 65, 65, 32, 1, False),
          (3, 30, 90, 32, 1, True),
          (2, 20, 100, 64, 1, False),
          (7, 78, 78, 32, 1, True)]    # staged width exactly 80


@pytest.mark.parametrize("B,H,W,C,stride,skip", SHAPES)
def test_dwconv_fwd_and_bwd_against_float64(B, H, W, C, stride, skip):
    import trackertraincode._hip as hip
    L, p = hip.lib(), hip.ptr
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W + C + stride)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    yprev, skp = rnd(B, H, W, C), (rnd(B, H, W, C).abs() if skip else None)
    w = rnd(C, 1, 3, 3) * 0.3
    bn_prev, bn_dw = _bn(C, g), _bn(C, g)
    f32 = lambda t: None if t is None else t.to(torch.float32).cuda().contiguous()
    # work with the float32-rounded inputs so that the reference sees exactly what the kernel sees
    yprev, w, bn_prev, bn_dw = (t.to(torch.float32).double() for t in (yprev, w, bn_prev, bn_dw))
    skp = None if skp is None else skp.to(torch.float32).double()

    # ---------------- forward
    pre = bn_prev[SCALE] * (yprev - bn_prev[MEAN]) + bn_prev[BETA] + (skp if skip else 0.0)
    a_in = pre.clamp_min(0.0)
    y_ref = F.conv2d(_nchw(a_in), w, stride=stride, padding=1, groups=C).permute(0, 2, 3, 1)
    blk = lambda t: None if t is None else hip.to_blocks(f32(t))  # activations travel as channel blocks (include/ttk.h)
    unblk = lambda t: hip.from_blocks(t).cpu().double()
    d_yprev, d_skip, d_w, d_bnp = blk(yprev), blk(skp), f32(w), f32(bn_prev)
    want_a = skip and stride == 1
    a_out = torch.full((B, H, W, C), float("nan"), device="cuda") if want_a else None
    y = torch.full((B, Ho, Wo, C), float("nan"), device="cuda")
    rows = L.partial_rows_dwconv(B, H, W, C, stride, False)
    part = torch.full((rows, 2, C), float("nan"), device="cuda")
    piv = (torch.randn(C, generator=torch.Generator().manual_seed(C)) * 0.5).float()  # statistics pivot (include/ttk.h)
    d_piv = piv.cuda()
    L.call("ttk_dwconv3x3_fwd", p(d_yprev), p(d_bnp), p(d_skip), p(a_out), p(d_w), p(y), p(part), p(d_piv), B, H, W, C, stride, 0)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(part).all()
    scale = y_ref.abs().max().item()
    assert (unblk(y) - y_ref).abs().max().item() <= 3e-6 * scale
    if want_a:
        assert (unblk(a_out) - a_in).abs().max().item() <= 1e-6 * max(a_in.abs().max().item(), 1.0)
    ps = part.cpu().double().sum(0)
    ys = y_ref - piv.double()
    assert torch.allclose(ps[0], ys.sum((0, 1, 2)), rtol=0, atol=2e-5 * ys.abs().sum((0, 1, 2)).max().item())
    assert torch.allclose(ps[1], (ys ** 2).sum((0, 1, 2)), rtol=2e-5, atol=1e-12)

    # ---------------- data gradient (+ fused weight gradient), block input recomputed and materialised
    g_dw, y_dw = rnd(B, Ho, Wo, C).to(torch.float32).double(), unblk(y)
    sg = rnd(B, H, W, C).to(torch.float32).double() if (skip and stride == 1) else None
    dy = bn_dw[GA] * (g_dw - bn_dw[GMEAN]) + bn_dw[GB] * (y_dw - bn_dw[MEAN])
    a_leaf = a_in.clone().requires_grad_(True)
    w_leaf = w.clone().requires_grad_(True)
    out = F.conv2d(_nchw(a_leaf), w_leaf, stride=stride, padding=1, groups=C)
    out.backward(_nchw(dy).contiguous())
    G = a_leaf.grad + (sg if sg is not None else 0.0)
    margin = pre.abs() > 1e-4  # entries whose relu mask could flip with rounding are left out
    gp_ref = G * (pre > 0)
    dw_ref = w_leaf.grad.reshape(C, 9)
    d_g, d_y, d_bnd, d_sg = blk(g_dw), y, f32(bn_dw), blk(sg)
    rows_b = L.partial_rows_dwconv(B, H, W, C, stride, True)
    for materialised in ((False, True) if want_a else (False,)):
        g_prev = torch.full((B, H, W, C), float("nan"), device="cuda")
        part_b = torch.full((rows_b, 2, C), float("nan"), device="cuda")
        dwg = torch.zeros(C, 9, device="cuda")
        L.call("ttk_dwconv3x3_bwd_data", p(d_g), p(d_y), p(d_bnd), p(d_w), p(d_sg), p(d_yprev), p(d_bnp), p(d_skip),
               p(a_out) if materialised else None, p(g_prev), p(part_b), p(dwg), 1, None, B, H, W, C, stride, 0)
        torch.cuda.synchronize()
        # deterministic form of the fused weight gradient: per-workgroup rows folded in a fixed order, twice bit-identical
        det = []
        for _ in range(2):
            scratch = torch.full((rows_b * 9 * C,), float("nan"), device="cuda")
            dwd, gp2, pb2 = torch.zeros(C, 9, device="cuda"), torch.empty_like(g_prev), torch.empty_like(part_b)
            L.call("ttk_dwconv3x3_bwd_data", p(d_g), p(d_y), p(d_bnd), p(d_w), p(d_sg), p(d_yprev), p(d_bnp), p(d_skip),
                   p(a_out) if materialised else None, p(gp2), p(pb2), p(dwd), 1, p(scratch), B, H, W, C, stride, 0)
            torch.cuda.synchronize()
            det.append(dwd)
        assert torch.equal(det[0], det[1])
        assert (det[0].cpu().double() - dw_ref).abs().max().item() <= 3e-5 * dw_ref.abs().max().item()
        got = unblk(g_prev)
        assert torch.isfinite(got).all() and torch.isfinite(part_b).all()
        sc = gp_ref.abs().max().item()
        assert ((got - gp_ref) * margin).abs().max().item() <= 5e-6 * sc, (materialised,)
        assert (dwg.cpu().double() - dw_ref).abs().max().item() <= 3e-5 * dw_ref.abs().max().item()
        pb = part_b.cpu().double().sum(0)
        assert torch.allclose(pb[0], got.sum((0, 1, 2)), rtol=0, atol=3e-5 * got.abs().sum((0, 1, 2)).max().item())
        s2 = (got * (yprev - bn_prev[MEAN])).sum((0, 1, 2))
        assert torch.allclose(pb[1], s2, rtol=0, atol=3e-5 * (got * (yprev - bn_prev[MEAN])).abs().sum((0, 1, 2)).max().item())


@pytest.mark.parametrize("B,H,W", [(2, 129, 129), (3, 33, 33), (2, 40, 50), (5, 7, 9)])
def test_stem_fwd_and_weight_gradient_against_float64(B, H, W):
    """5x5/s2 stem on the fp32 matrix cores (csrc/stem.hip) against torch float64, odd/even/small sizes."""
    import trackertraincode._hip as hip
    L, p = hip.lib(), hip.ptr
    g = torch.Generator().manual_seed(B + H + W)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    x = (torch.rand(B, 1, H, W, generator=g) - 0.5).double()
    w = (torch.randn(32, 1, 5, 5, generator=g) * 0.2).float().double()
    x = x.float().double()
    y_ref = F.conv2d(x, w, stride=2, padding=2).permute(0, 2, 3, 1)
    d_x, d_w = x.float().cuda(), w.float().cuda()
    y = torch.full((B, Ho, Wo, 32), float("nan"), device="cuda")
    rows = L.partial_rows_elementwise(B * Ho * Wo * 8)
    part = torch.full((rows, 2, 32), float("nan"), device="cuda")
    piv = (torch.randn(32, generator=g) * 0.1).float()  # statistics pivot (include/ttk.h)
    d_piv = piv.cuda()
    L.call("ttk_stem_fwd", p(d_x), p(d_w), p(y), p(part), p(d_piv), B, H, W, 0)
    torch.cuda.synchronize()
    assert (y.cpu().double() - y_ref).abs().max().item() <= 3e-6 * y_ref.abs().max().item()
    ps = part.cpu().double().sum(0)
    ys = y_ref - piv.double()
    assert torch.allclose(ps[0], ys.sum((0, 1, 2)), rtol=0, atol=2e-5 * ys.abs().sum((0, 1, 2)).max().item())
    assert torch.allclose(ps[1], (ys ** 2).sum((0, 1, 2)), rtol=2e-5)
    bn = _bn(32, g).float().double()
    gr = torch.randn(B, Ho, Wo, 32, generator=g).float().double()
    yv = y.cpu().double()
    dy = bn[GA] * (gr - bn[GMEAN]) + bn[GB] * (yv - bn[MEAN])
    wl = w.clone().requires_grad_(True)
    F.conv2d(x, wl, stride=2, padding=2).backward(_nchw(dy).contiguous())
    dw = torch.zeros(32, 25, device="cuda")
    d_g, d_bn = gr.float().cuda(), bn.float().cuda()
    L.call("ttk_stem_bwd_weight", p(d_g), p(y), p(d_bn), p(d_x), p(dw), 1, None, B, H, W, 0)
    torch.cuda.synchronize()
    ref = wl.grad.reshape(32, 25)
    assert (dw.cpu().double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    det = []
    for _ in range(2):  # deterministic form: workgroup partials in scratch, folded in a fixed order
        scratch = torch.full((L.cdll.ttk_stem_wgrad_partial_bytes() // 4,), float("nan"), device="cuda")
        dwd = torch.zeros(32, 25, device="cuda")
        L.call("ttk_stem_bwd_weight", p(d_g), p(y), p(d_bn), p(d_x), p(dwd), 1, p(scratch), B, H, W, 0)
        torch.cuda.synchronize()
        det.append(dwd)
    assert torch.equal(det[0], det[1])
    assert (det[0].cpu().double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
