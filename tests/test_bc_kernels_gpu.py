"""Kernels of the bf16-COMPUTE path (csrc/bc_*.hip, entry points ttk_bc_* of include/ttk.h) against float64 torch references that see exactly
the kernels' operands: bf16-stored tensors, the one-fma BatchNorm maps, operands rounded to bf16 before the product, fp32 accumulation.

What is compared with what (tolerances written at the assertions):
  * outputs (bf16) against the reference rounded to bf16: at most one bf16 step (2^-8 relative) where the two fp32 sums round to different
    sides, and an absolute floor for cancelled sums;
  * BatchNorm partial sums against sums of the kernel's OWN stored output (they must describe what the consumer will read): fp32 summation
    error only;
  * weight gradients (fp32) against the float64 contraction of the same bf16 operands: 1e-4 of the tensor's scale.
The reference trains in fp32 only (scripts/train_poseestimator.py:442-454): this mode has no reference counterpart, its whole-step parity
against the fp32 oracle is in tests/test_bf16_compute_gpu.py.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SCALE, BETA, MEAN, RSTD, GA, GB, GMEAN = range(7)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _bn(C, g):
    bn = torch.zeros(8, C, dtype=torch.float64)
    bn[SCALE] = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    bn[BETA] = torch.randn(C, generator=g, dtype=torch.float64) * 0.2
    bn[MEAN] = torch.randn(C, generator=g, dtype=torch.float64) * 0.3
    bn[RSTD] = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    bn[GA] = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    bn[GB] = torch.randn(C, generator=g, dtype=torch.float64) * 0.2
    bn[GMEAN] = torch.randn(C, generator=g, dtype=torch.float64) * 0.05
    return bn.to(torch.float32).double()


def _fwd_map(y, bn):  # scale*y + shift in fp32 (the kernels: one fma, shift = fma(-scale, mean, beta)); returned as float64
    sc = bn[SCALE].float()
    sh = bn[BETA].float() - sc * bn[MEAN].float()
    return (sc * y.float() + sh).double()


def _bwd_map(g, y, bn):  # ga*g + gb*y + c0 in fp32
    ga, gb = bn[GA].float(), bn[GB].float()
    c0 = -ga * bn[GMEAN].float() - gb * bn[MEAN].float()
    return (ga * g.float() + (gb * y.float() + c0)).double()


def _close_bf16(got, ref, what, steps=1.0, floor=0.5):
    """`got` (bf16 values as float64) against `ref` (float64, unrounded): |got - ref| <= steps * 2^-8 * |ref| + floor * 2^-8 * rms(ref).
    A few elements per million may miss it: where an operand sits on a bf16 rounding tie (or a ReLU input within rounding of zero) the
    kernel's fp32 arithmetic and the reference's float64 arithmetic round it to different sides - one mask decision or one operand step."""
    tol = steps * 2.0 ** -8 * ref.abs() + floor * 2.0 ** -8 * ref.pow(2).mean().sqrt()
    bad = (got - ref).abs() > tol
    assert int(bad.sum()) <= 1e-5 * bad.numel(), f"{what}: {int(bad.sum())} of {bad.numel()} off; worst {(got - ref).abs().max().item():.3e} at ref {ref[bad][0].item():.3e}"


def _dev(t, dtype):
    return t.to(dtype).cuda().contiguous()


PW_SHAPES = [(1000, 32, 64), (777, 64, 128), (1500, 128, 128), (900, 128, 256), (1300, 256, 256), (700, 256, 512), (1111, 512, 512),
             (300, 512, 1024), (520, 1024, 1024), (41, 512, 512), (20000, 32, 64), (70001, 64, 128), (40000, 128, 128), (33, 32, 64), (1, 128, 128)]


@pytest.mark.parametrize("M,cin,cout", PW_SHAPES)
def test_pointwise_forward_datagrad_weightgrad(M, cin, cout):
    import trackertraincode._hip as hip
    L, p = hip.lib(), hip.ptr
    g = torch.Generator().manual_seed(M + cin * 7 + cout)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    w = (rnd(cout, cin) / cin ** 0.5).to(torch.float32).double()
    ydw = bf(rnd(M, cin))
    bn_dw, bn_pw = _bn(cin, g), _bn(cout, g)
    prep = torch.empty(L.cdll.ttk_bc_prepared_bytes(cin, cout), dtype=torch.uint8, device="cuda")
    d_w = _dev(w.view(cout, cin, 1, 1), torch.float32)
    L.bc_prepare_weights([d_w], [prep])
    wb = bf(w)
    # ---------------- forward
    a = bf(_fwd_map(ydw, bn_dw)).clamp_min(0.0)
    y_ref = a @ wb.T
    d_ydw = hip.to_blocks64(_dev(ydw, torch.bfloat16))
    d_bn_dw, d_bn_pw = _dev(bn_dw, torch.float32), _dev(bn_pw, torch.float32)
    y = torch.full((M, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    rows = L.cdll.ttk_bc_partial_rows_pw(M, cin, cout)
    assert rows > 0
    part = torch.full((rows, 2, cout), float("nan"), device="cuda")
    piv = (torch.randn(cout, generator=g) * 0.3).float()
    L.call("ttk_bc_pw_fwd", p(d_ydw), p(d_bn_dw), p(prep), p(y), p(part), p(piv.cuda()), M, cin, cout)
    torch.cuda.synchronize()
    y_got = hip.from_blocks64(y).cpu().double()
    assert torch.isfinite(y_got).all()
    _close_bf16(y_got, y_ref, "forward output")
    s = part.double().sum(0).cpu()
    d = y_got - piv.double()
    assert torch.allclose(s[0], d.sum(0), rtol=1e-4, atol=1e-3 + 1e-6 * d.abs().sum(0).max().item()), "forward partial sums (1)"
    assert torch.allclose(s[1], (d * d).sum(0), rtol=1e-4, atol=1e-3), "forward partial sums (2)"
    # ---------------- data gradient
    gy, yy = bf(rnd(M, cout) * 0.1), bf(rnd(M, cout))
    dy = bf(_bwd_map(gy, yy, bn_pw))
    mask = _fwd_map(ydw, bn_dw).float() > 0  # the kernel's mask: the fp32 map's sign
    gd_ref = (dy @ wb) * mask
    d_g, d_y = hip.to_blocks64(_dev(gy, torch.bfloat16)), hip.to_blocks64(_dev(yy, torch.bfloat16))
    g_dw = torch.full((M, cin), float("nan"), dtype=torch.bfloat16, device="cuda")
    rows = L.cdll.ttk_bc_partial_rows_pw(M, cout, cin)
    part = torch.full((rows, 2, cin), float("nan"), device="cuda")
    L.call("ttk_bc_pw_bwd_data", p(d_g), p(d_y), p(d_bn_pw), p(prep), p(d_ydw), p(d_bn_dw), p(g_dw), p(part), M, cin, cout)
    torch.cuda.synchronize()
    gd_got = hip.from_blocks64(g_dw).cpu().double()
    assert torch.isfinite(gd_got).all()
    # elements whose fp32 map is within rounding of zero may take either side of the mask: leave them out
    near = _fwd_map(ydw, bn_dw).abs() < 1e-5
    _close_bf16(torch.where(near, gd_ref, gd_got), gd_ref, "data gradient")
    s = part.double().sum(0).cpu()
    assert torch.allclose(s[0], gd_got.sum(0), rtol=1e-4, atol=1e-4), "data-gradient partial sums (1)"
    assert torch.allclose(s[1], (gd_got * (ydw - bn_dw[MEAN])).sum(0), rtol=1e-4, atol=1e-3), "data-gradient partial sums (2)"
    # ---------------- weight gradient (accumulates into dw)
    dw0 = rnd(cout, cin).float()
    dw = dw0.clone().cuda()
    scratch = torch.empty(L.cdll.ttk_bc_pw_wgrad_scratch_bytes(M, cin, cout) // 4, dtype=torch.float32, device="cuda")
    L.call("ttk_bc_pw_bwd_weight", p(d_g), p(d_y), p(d_bn_pw), p(d_ydw), p(d_bn_dw), p(dw), p(scratch), M, cin, cout)
    torch.cuda.synchronize()
    dw_ref = dy.T @ a
    err = (dw.cpu().double() - dw0.double() - dw_ref).abs().max().item()
    assert err <= 1e-4 * dw_ref.abs().max().item() + 1e-5, f"weight gradient off by {err:.3e} (scale {dw_ref.abs().max().item():.3e})"
    # ---------------- both in one kernel (the early layers)
    rows = L.cdll.ttk_bc_pw_bwd_fused_rows(M, cin, cout)
    assert (rows > 0) == ((cin, cout) in ((32, 64), (64, 128), (128, 128)))
    if rows > 0:
        g_dw2 = torch.full((M, cin), float("nan"), dtype=torch.bfloat16, device="cuda")
        part2 = torch.full((rows, 2, cin), float("nan"), device="cuda")
        dw2 = dw0.clone().cuda()
        scratch = torch.full((L.cdll.ttk_bc_pw_bwd_fused_scratch_bytes(M, cin, cout) // 4,), float("nan"), dtype=torch.float32, device="cuda")
        L.call("ttk_bc_pw_bwd_fused", p(d_g), p(d_y), p(d_bn_pw), p(prep), p(d_ydw), p(d_bn_dw), p(g_dw2), p(dw2), p(scratch), p(part2), M, cin, cout)
        torch.cuda.synchronize()
        assert torch.equal(g_dw2.view(torch.int16), g_dw.view(torch.int16)), "fused data gradient differs from ttk_bc_pw_bwd_data's"
        s2 = part2.double().sum(0).cpu()
        assert torch.allclose(s2[0], gd_got.sum(0), rtol=1e-4, atol=1e-4) and torch.allclose(s2[1], (gd_got * (ydw - bn_dw[MEAN])).sum(0), rtol=1e-4, atol=1e-3)
        err = (dw2.cpu().double() - dw0.double() - dw_ref).abs().max().item()
        assert err <= 1e-4 * dw_ref.abs().max().item() + 1e-5, f"fused weight gradient off by {err:.3e}"
    # ---------------- deferred fold: dw = NULL leaves the slice tiles in scratch; ttk_bc_bn_bwd_finalize_fold adds them in the launch that finalises
    # bn_dw's backward - bitwise what the separate launches give (weight gradient, BatchNorm-backward constants, dgamma / dbeta)
    gamma = (torch.rand(cin, generator=g) + 0.5).float().cuda()
    rows_d = L.cdll.ttk_bc_partial_rows_pw(M, cout, cin)
    for fused in ((False, True) if rows > 0 else (False,)):
        outs = []
        for defer in (False, True):
            dwx, bnx = dw0.clone().cuda(), d_bn_dw.clone()
            dgm, dbt = torch.zeros(cin, device="cuda"), torch.zeros(cin, device="cuda")
            if fused:
                n_rows = slices = rows
                scr = torch.full((L.cdll.ttk_bc_pw_bwd_fused_scratch_bytes(M, cin, cout) // 4,), float("nan"), dtype=torch.float32, device="cuda")
                prt = torch.full((n_rows, 2, cin), float("nan"), device="cuda")
                L.call("ttk_bc_pw_bwd_fused", p(d_g), p(d_y), p(d_bn_pw), p(prep), p(d_ydw), p(d_bn_dw), p(g_dw), None if defer else p(dwx), p(scr), p(prt), M, cin, cout)
            else:
                n_rows, slices = rows_d, L.cdll.ttk_bc_pw_wgrad_slices(M, cin, cout)
                assert slices > 0
                scr = torch.full((L.cdll.ttk_bc_pw_wgrad_scratch_bytes(M, cin, cout) // 4,), float("nan"), dtype=torch.float32, device="cuda")
                prt = torch.full((n_rows, 2, cin), float("nan"), device="cuda")
                L.call("ttk_bc_pw_bwd_weight", p(d_g), p(d_y), p(d_bn_pw), p(d_ydw), p(d_bn_dw), None if defer else p(dwx), p(scr), M, cin, cout)
                L.call("ttk_bc_pw_bwd_data", p(d_g), p(d_y), p(d_bn_pw), p(prep), p(d_ydw), p(d_bn_dw), p(g_dw), p(prt), M, cin, cout)
            if defer:
                L.call("ttk_bc_bn_bwd_finalize_fold", p(prt), n_rows, cin, M, p(gamma), p(bnx), p(dgm), p(dbt), 0, p(scr), slices, cin * cout, p(dwx), 1)
            else:
                L.call("ttk_bn_bwd_finalize", p(prt), n_rows, cin, M, p(gamma), p(bnx), p(dgm), p(dbt), 0)
            torch.cuda.synchronize()
            outs.append((dwx.clone(), bnx.clone(), dgm.clone(), dbt.clone()))
        for a_, b_ in zip(*outs):
            assert torch.equal(a_, b_), f"deferred fold differs from the separate launches (fused={fused})"
        err = (outs[1][0].cpu().double() - dw0.double() - dw_ref).abs().max().item()
        assert err <= 1e-4 * dw_ref.abs().max().item() + 1e-5, f"deferred weight gradient off by {err:.3e}"


def _nchw(t):
    return t.permute(0, 3, 1, 2)


# (B, H, W, C, stride, skip)
DW_SHAPES = [(4, 9, 9, 64, 1, True), (5, 5, 5, 128, 1, False), (9, 5, 5, 64, 1, True), (2, 50, 50, 32, 1, False), (2, 40, 70, 64, 1, True),
             (2, 65, 65, 32, 1, False), (3, 33, 33, 64, 1, True), (2, 33, 33, 64, 2, False), (3, 10, 12, 64, 2, False), (6, 9, 9, 128, 2, False),
             (1, 17, 17, 256, 1, True), (40, 65, 65, 32, 1, True), (40, 65, 65, 64, 2, False), (100, 33, 33, 64, 1, True), (90, 33, 33, 128, 2, False)]


@pytest.mark.parametrize("B,H,W,C,stride,skip", DW_SHAPES)
def test_depthwise_forward_and_backward(B, H, W, C, stride, skip):
    import trackertraincode._hip as hip
    L, p = hip.lib(), hip.ptr
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W + C + stride)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    yprev, skp = bf(rnd(B, H, W, C)), (bf(rnd(B, H, W, C).abs()) if skip else None)
    w = (rnd(C, 1, 3, 3) * 0.3).to(torch.float32).double()
    bn_prev, bn_dw = _bn(C, g), _bn(C, g)
    blk = lambda t: None if t is None else hip.to_blocks64(_dev(t, torch.bfloat16))
    unblk = lambda t: hip.from_blocks64(t).cpu().double()
    # ---------------- forward: a_in = relu(bf16(scale*y + shift (+ skip))), conv in fp32 on the bf16 tile
    a_in = bf(_fwd_map(yprev, bn_prev) + (skp if skip else 0.0)).clamp_min(0.0)
    y_ref = F.conv2d(_nchw(a_in), w, stride=stride, padding=1, groups=C).permute(0, 2, 3, 1)
    d_yprev, d_skip, d_w, d_bnp, d_bnd = blk(yprev), blk(skp), _dev(w, torch.float32), _dev(bn_prev, torch.float32), _dev(bn_dw, torch.float32)
    want_a = skip and stride == 1
    a_out = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device="cuda") if want_a else None
    y = torch.full((B, Ho, Wo, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    rows = L.cdll.ttk_bc_partial_rows_dw(B, H, W, C, stride, 0)
    part = torch.full((rows, 2, C), float("nan"), device="cuda")
    piv = (torch.randn(C, generator=g) * 0.5).float()
    L.call("ttk_bc_dw_fwd", p(d_yprev), p(d_bnp), p(d_skip), p(a_out), p(d_w), p(y), p(part), p(piv.cuda()), B, H, W, C, stride)
    torch.cuda.synchronize()
    y_got = unblk(y)
    assert torch.isfinite(y_got).all()
    _close_bf16(y_got, y_ref, "depthwise forward")
    if want_a:
        assert torch.equal(unblk(a_out), a_in) or (unblk(a_out) - a_in).abs().max() <= 2.0 ** -8 * a_in.abs().max(), "materialised block input"
    s = part.double().sum(0).cpu()
    d = y_got - piv.double()
    assert torch.allclose(s[0], d.sum((0, 1, 2)), rtol=1e-4, atol=1e-3), "forward partial sums (1)"
    assert torch.allclose(s[1], (d * d).sum((0, 1, 2)), rtol=1e-4, atol=1e-3), "forward partial sums (2)"
    # ---------------- backward
    ydw = bf(y_ref)
    gdw = bf(rnd(B, Ho, Wo, C) * 0.1)
    sg = bf(rnd(B, H, W, C) * 0.1) if want_a else None
    dy = bf(_bwd_map(gdw, ydw, bn_dw))
    G = F.conv_transpose2d(_nchw(dy), w, stride=stride, padding=1, groups=C, output_padding=((H + 2 - 3) % stride, (W + 2 - 3) % stride)).permute(0, 2, 3, 1)
    if sg is not None:
        G = G + sg
    gp_ref = G * (a_in > 0)
    # fused depthwise weight gradient: sum over pixels of dy (taps) * a_in
    xin = _nchw(a_in).reshape(1, B * C, H, W)
    dw_ref = F.conv2d(xin, _nchw(dy).reshape(B * C, 1, Ho, Wo), padding=1, dilation=stride, groups=B * C)[:, :, :3, :3].reshape(B, C, 3, 3).sum(0)
    d_g, d_yd, d_sg = blk(gdw), blk(ydw), blk(sg)
    g_prev = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    rows = L.cdll.ttk_bc_partial_rows_dw(B, H, W, C, stride, 1)
    part = torch.full((rows, 2, C), float("nan"), device="cuda")
    dw = torch.zeros(C, 1, 3, 3, device="cuda")
    for use_a in ((True, False) if want_a else (False,)):
        dw.zero_()
        L.call("ttk_bc_dw_bwd_data", p(d_g), p(d_yd), p(d_bnd), p(d_w), p(d_sg), p(d_yprev), p(d_bnp), p(d_skip), p(a_out) if use_a else None, p(g_prev),
               p(part), p(dw), 1, None, B, H, W, C, stride)
        torch.cuda.synchronize()
        gp_got = unblk(g_prev)
        assert torch.isfinite(gp_got).all()
        _close_bf16(gp_got, gp_ref, f"depthwise data gradient (a_in {'given' if use_a else 'recomputed'})")
        err = (dw.cpu().double().view(C, 3, 3) - dw_ref).abs().max().item()
        assert err <= 2e-4 * dw_ref.abs().max().item() + 1e-4, f"fused depthwise weight gradient off by {err:.3e} (scale {dw_ref.abs().max().item():.3e})"
        s = part.double().sum(0).cpu()
        assert torch.allclose(s[0], gp_got.sum((0, 1, 2)), rtol=1e-4, atol=1e-3), "backward partial sums (1)"
        assert torch.allclose(s[1], (gp_got * (yprev - bn_prev[MEAN])).sum((0, 1, 2)), rtol=1e-4, atol=2e-3), "backward partial sums (2)"
    # ---------------- workgroup rows + ONE launch for "fold the rows" and "finalise the producer's BatchNorm backward" = the two separate calls, bitwise
    scr = torch.full((rows * 9 * C,), float("nan"), device="cuda")
    gamma = (torch.rand(C, generator=g) + 0.5).float().cuda()
    outs = []
    for fused_launch in (False, True):
        dwx, bnx = torch.full((C, 1, 3, 3), 0.25, device="cuda"), d_bnp.clone()
        dgm, dbt = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        L.call("ttk_bc_dw_bwd_data", p(d_g), p(d_yd), p(d_bnd), p(d_w), p(d_sg), p(d_yprev), p(d_bnp), p(d_skip), p(a_out), p(g_prev), p(part), p(dwx),
               2 if fused_launch else 1, p(scr), B, H, W, C, stride)
        if fused_launch:
            L.call("ttk_bc_bn_bwd_finalize_fold", p(part), rows, C, B * H * W, p(gamma), p(bnx), p(dgm), p(dbt), 0, p(scr), rows, 9 * C, p(dwx), 1)
        else:
            L.call("ttk_bn_bwd_finalize", p(part), rows, C, B * H * W, p(gamma), p(bnx), p(dgm), p(dbt), 0)
        torch.cuda.synchronize()
        outs.append((dwx.clone(), bnx.clone(), dgm.clone(), dbt.clone()))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_), "combined finalisation + fold launch differs from the two separate launches"
    err = (outs[1][0].cpu().double().view(C, 3, 3) - 0.25 - dw_ref).abs().max().item()
    assert err <= 2e-4 * dw_ref.abs().max().item() + 1e-4, f"weight gradient through the row fold off by {err:.3e}"
