"""BatchNorm statistics of channels whose mean lies many standard deviations from zero (include/ttk.h, "the statistics PIVOT").

The forward producers leave fp32 partial sums; formed as sum(y) and sum(y^2) they lose ~ (mean/sigma)^2 * 2^-24 of the variance of
such a channel to cancellation (13 sigma: 1e-5 relative - visible in running_var against torch).  With the layer's running mean as
the pivot the producers sum (y - pivot) and (y - pivot)^2 and the finalisation adds the pivot back.  Here every producer writes a
tensor with channels at ~13 sigma, the pivot is a running mean that earlier batches of the same distribution would have left
(the batch mean +- a fraction of sigma), and what ttk_bn_fwd_finalize then writes - the BatchNorm rows and the updated running
statistics - is compared with float64 statistics of the very tensor the kernel stored (reference: F.batch_norm(training=True)
through nn.BatchNorm2d, backbones/mobilenet_v1.py:30,66,68).  Bar: 1e-6 relative in running_var (fp32 storage of the result itself
is 6e-8)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BN_SCALE, BN_BETA, BN_MEAN, BN_RSTD, BN_GA, BN_GB, BN_GMEAN, BN_AUX = range(8)
MOMENTUM, EPS = 0.1, 1e-5


def _H():
    import trackertraincode._hip as H
    return H


def _finalize_and_check(L, p, y, part_fn, C, label, rows_layout=False):
    """`part_fn(pivot_or_None)` runs the producer and returns its partial rows; `y` is what it stored ([..., C], on the GPU)."""
    dev = y.device
    y64 = (y if rows_layout else _H().from_blocks(y)).detach().double().reshape(-1, C).cpu()
    n = y64.shape[0]
    mean, var = y64.mean(0), y64.var(0, unbiased=False)
    ratio = (mean.abs() / var.sqrt()).numpy()
    assert (ratio > 10).sum() >= 1, f"{label}: the construction should put channels at >= 10 sigma, got max {ratio.max():.1f}"
    gen = torch.Generator().manual_seed(C)
    gamma, beta = (torch.rand(C, generator=gen) + 0.5).to(dev), (torch.randn(C, generator=gen) * 0.2).to(dev)
    rv0 = (torch.rand(C, generator=gen) + 0.5)
    rm0 = (mean + 0.3 * var.sqrt() * torch.randn(C, generator=gen, dtype=torch.float64)).float()  # what earlier batches left behind
    want_rm = (1 - MOMENTUM) * rm0.double() + MOMENTUM * mean
    want_rv = (1 - MOMENTUM) * rv0.double() + MOMENTUM * var * n / (n - 1)
    want_rstd = 1.0 / torch.sqrt(var + EPS)
    errs = {}
    for use_pivot in (True, False):
        rm, rv = rm0.clone().to(dev), rv0.clone().to(dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        bn = torch.zeros(8, C, device=dev)
        part = part_fn(rm if use_pivot else None)
        L.call("ttk_bn_fwd_finalize", p(part), p(rm) if use_pivot else None, part.shape[0], C, n, p(gamma), p(beta), p(rm), p(rv), p(nbt),
               MOMENTUM, EPS, p(bn))
        torch.cuda.synchronize()
        rel = lambda got, want: float(((got.double().cpu() - want).abs() / want.abs().clamp_min(1e-30)).max())
        rel_m = lambda got, want: float(((got.double().cpu() - want).abs() / torch.maximum(want.abs(), var.sqrt())).max())  # means near zero: in sigma
        errs[use_pivot] = dict(running_var=rel(rv, want_rv), running_mean=rel_m(rm, want_rm), rstd=rel(bn[BN_RSTD], want_rstd),
                               mean=rel_m(bn[BN_MEAN], mean), scale=rel(bn[BN_SCALE], gamma.double().cpu() * want_rstd))
        assert int(nbt) == 1
    print(f"{label}: max |mean|/sigma {ratio.max():.1f};  with pivot {errs[True]};  plain sums {errs[False]}")
    for k, v in errs[True].items():
        assert v < 1e-6, (label, k, v, errs)
    return errs


def _lib():
    import trackertraincode._hip as H
    return H.lib(), H.ptr


# (M, Cin, Cout): row-block kernel (csrc/pwconv_r.hip), 128 x 256 / 256 x 128 tiles of csrc/pwconv_f16.hip, fp32-MFMA shapes of csrc/pwconv.hip
@pytest.mark.parametrize("M,Cin,Cout", [(20736, 512, 512), (8192, 128, 256), (4100, 256, 128), (5000, 32, 64), (3000, 64, 128)])
def test_pointwise_producer(M, Cin, Cout):
    L, p = _lib()
    rng = np.random.default_rng(M + Cout)
    dev = "cuda"
    ydw = rng.normal(0, 1, (M, Cin)).astype(np.float32)
    bn_dw = np.zeros((8, Cin), np.float32)
    bn_dw[BN_SCALE], bn_dw[BN_BETA] = 1.0, 0.0
    bn_dw[BN_SCALE, 0], bn_dw[BN_BETA, 0] = 0.0, 1.0           # input channel 0 is the constant 1 ...
    w = (rng.normal(0, 1, (Cout, Cin)) / np.sqrt(Cin)).astype(np.float32)
    sigma = np.sqrt(0.34 * (w[:, 1:] ** 2).sum(1))              # (var of relu(z) = 0.34)
    w[:, 0] = np.where(np.arange(Cout) % 3 == 0, 13.0 * sigma, w[:, 0])  # ... which lifts every third output channel to ~13 sigma
    a_max = float(np.maximum(ydw, 0).max())
    bn_dw[BN_AUX, 0] = max(a_max, 1.0)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_ydw, d_bn, d_w = _H().to_blocks(t(ydw)), t(bn_dw), t(w)
    y = torch.empty(M, Cout, device=dev)
    wq = torch.empty(L.pwconv_prepared_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
    rows = L.partial_rows_gemm(M, Cin, Cout)

    def run(pivot):
        part = torch.full((rows, 2, Cout), float("nan"), device=dev)
        L.call("ttk_pwconv1x1_fwd", p(d_ydw), p(d_bn), p(d_w), p(y), p(part), p(pivot), M, Cin, Cout, p(wq), 0)
        return part

    run(None)
    _finalize_and_check(L, p, y, run, Cout, f"pwconv M={M} {Cin}->{Cout}")


# several images per tile / row bands, stride 2, the 512-channel slabs
@pytest.mark.parametrize("B,H,C,stride", [(64, 17, 64, 1), (32, 33, 128, 2), (256, 9, 512, 1)])
def test_depthwise_producer(B, H, C, stride):
    """A large constant input with small noise under filters that only have their centre tap (with more taps the zero padding makes
    the border pixels smaller and that spread, not the noise, sets sigma)."""
    L, p = _lib()
    g = torch.Generator().manual_seed(B + C)
    dev = "cuda"
    yprev = torch.randn(B, H, H, C, generator=g)
    bn_prev = torch.zeros(8, C)
    bn_prev[BN_SCALE], bn_prev[BN_BETA] = 0.2, 2.6          # a_in = 2.6 + 0.2 z: 13 sigma
    w = torch.zeros(C, 1, 3, 3)
    w[:, 0, 1, 1] = torch.rand(C, generator=g) + 0.5       # y = w * a_in
    Ho = (H - 1) // stride + 1
    d_yprev, d_bn, d_w = _H().to_blocks(yprev.to(dev)), bn_prev.to(dev), w.to(dev)
    y = torch.empty(B, Ho, Ho, C, device=dev)
    rows = L.partial_rows_dwconv(B, H, H, C, stride, False)

    def run(pivot):
        part = torch.full((rows, 2, C), float("nan"), device=dev)
        L.call("ttk_dwconv3x3_fwd", p(d_yprev), p(d_bn), None, None, p(d_w), p(y), p(part), p(pivot), B, H, H, C, stride, 0)
        return part

    run(None)
    _finalize_and_check(L, p, y, run, C, f"dwconv B={B} {H}x{H} C={C} s{stride}")


def test_resnet_conv_and_stem_producers():
    L, p = _lib()
    rng = np.random.default_rng(3)
    dev = "cuda"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # 3 x 3 conv 64 -> 64: input = constant channel 0 + noise elsewhere, centre tap of channel 0 carries the offset
    B, H, Cin, Cout = 32, 17, 64, 64
    a = np.abs(rng.normal(0, 1, (B, H, H, Cin))).astype(np.float32)
    a[..., 0] = 1.0
    w = (rng.normal(0, 1, (Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    w[:, 0] = 0.0
    w[:, 0, 1, 1] = 8.0
    d_a, a_bound = t(a), t(np.array([np.abs(a).max()], np.float32))
    d_w = t(w)
    w_f = torch.empty(3, 9, Cout, Cin, dtype=torch.int16, device=dev)  # pre-split weight operands (tests/test_conv_gpu.py)
    w_b = torch.empty(3, 9, Cin, Cout, dtype=torch.int16, device=dev)
    L.call("ttk_conv_weight_repack", p(d_w), p(w_f), p(w_b), Cout, Cin, 3, 3)
    y = torch.empty(B, H, H, Cout, device=dev)
    rows = L.partial_rows_gemm(B * H * H)

    def run(pivot):
        part = torch.full((rows, 2, Cout), float("nan"), device=dev)
        L.call("ttk_conv_fwd", p(d_a), p(a_bound), p(w_f), p(y), p(part), p(pivot), B, H, H, Cin, Cout, 3, 3, 1, 1)
        return part

    run(None)
    _finalize_and_check(L, p, y, run, Cout, "conv3x3 64->64", rows_layout=True)

    # the two stems: a bright image (0.8 + noise) under filters whose taps share a sign
    for name, C, k, args in (("ttk_stem_fwd", 32, 5, (0,)), ("ttk_stem7_fwd", 64, 7, ())):
        Bs, Hs = 16, 129
        x = (0.8 + 0.02 * rng.normal(0, 1, (Bs, 1, Hs, Hs))).astype(np.float32)
        ws = (0.05 + 0.01 * rng.normal(0, 1, (C, 1, k, k))).astype(np.float32)
        d_x, d_ws = t(x), t(ws)
        Ho = (Hs + 1) // 2
        ys = torch.empty(Bs, Ho, Ho, C, device=dev)
        rows_s = L.partial_rows_elementwise(Bs * Ho * Ho * (C // 4))

        def run_s(pivot):
            part = torch.full((rows_s, 2, C), float("nan"), device=dev)
            L.call(name, p(d_x), p(d_ws), p(ys), p(part), p(pivot), Bs, Hs, Hs, *args)
            return part

        run_s(None)
        y64 = ys.double().reshape(-1, C).cpu()
        if float((y64.mean(0).abs() / y64.std(0)).max()) <= 10:  # the zero padding at the border dominates sigma
            print(name, "border-dominated sigma: ratio", float((y64.mean(0).abs() / y64.std(0)).max()))
            continue
        _finalize_and_check(L, p, ys, run_s, C, name, rows_layout=True)


def test_backbone_running_statistics_track_the_float64_oracle_over_steps():
    """The whole MobileNet backbone, six training-mode forwards from the initial running statistics over bright inputs (the first
    layers then carry channels far from zero): the running statistics of EVERY BatchNorm against the oracle evaluated in float64,
    held to what the oracle evaluated in float32 (the reference's own arithmetic) achieves."""
    from oracle import refmodel as R
    from oracle.synth import make_state
    from trackertraincode.backbones.mobilenet_v1 import MobileNet

    shapes = {k: v for k, v in R.state_shapes(False, False).items() if k.startswith("convnet.")}
    sd = make_state(shapes, 0)
    net = MobileNet(num_classes=None).cuda()
    net.load_state_dict({k[len("convnet."):]: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    net.train()

    def state(dtype):
        st = {}
        for k, v in sd.items():
            t = torch.from_numpy(np.array(v))
            st[k] = t.to(dtype) if t.is_floating_point() else t
        return st

    st64, st32 = state(torch.float64), state(torch.float32)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for step in range(6):
            x = torch.randn(12, 1, 129, 129, generator=g) * 0.3 + 1.5
            net.forward_features(x.cuda())
            R.mobilenet_forward(st64, x.double(), True)
            R.mobilenet_forward(st32, x, True)
    torch.cuda.synchronize()
    worst_hip = worst_cpu = 0.0
    for k, v in net.state_dict().items():
        if "running_" not in k:
            continue
        want = st64["convnet." + k]
        sigma = st64["convnet." + k.replace("running_mean", "running_var")].sqrt()
        scale = want.abs() if k.endswith("running_var") else sigma  # var: relative; mean: in standard deviations
        e_hip = float(((v.double().cpu() - want).abs() / scale).max())
        e_cpu = float(((st32["convnet." + k].double() - want).abs() / scale).max())
        worst_hip, worst_cpu = max(worst_hip, e_hip), max(worst_cpu, e_cpu)
        assert e_hip <= 3 * e_cpu + 2e-6, (k, e_hip, e_cpu)
    print(f"running statistics after six steps: worst deviation from float64  HIP {worst_hip:.2e}  fp32 oracle {worst_cpu:.2e}")
