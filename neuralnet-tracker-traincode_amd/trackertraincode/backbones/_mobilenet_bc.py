"""Forward / backward launch sequences of the MobileNet backbone on the bf16-COMPUTE path (`--precision bf16-compute`, BASELINE
config 5's bf16 leg; kernels csrc/bc_*.hip, entry points `ttk_bc_*` of include/ttk.h).

Same structure as `mobilenet_v1._forward_impl / _backward_impl` (reference `backbones/mobilenet_v1.py:96-189`: stem, 13 depthwise-separable
blocks, global average pool; BatchNorm in training mode split into producer partial sums, finalisation and apply-on-load), with

* every activation-sized tensor and its gradient in bfloat16, channel blocks of 64 (`_hip.to_blocks64`);
* the pointwise convolutions as ONE bf16 MFMA product with fp32 accumulation (no fp16 split, no operand bounds, no conversion waves);
* depthwise kernels with bf16 LDS tiles and fp32 accumulation;
* fp32 master weights, BatchNorm statistics, reductions, weight gradients and optimiser - unchanged.

The reference trains in fp32 only; this mode has no reference counterpart and its parity is stated as measured tolerances against the
fp32 oracle (tests/test_bf16_compute_gpu.py).  It is never the default and never the headline line of bench.py.
"""
from __future__ import annotations

import torch

from .. import _hip

_BF = 3          # TTK_STORE_ACT_BF16 | TTK_STORE_GRAD_BF16 (stem: C = 32, the same bytes in either block size)
_DT = torch.bfloat16


# Which weight-gradient folds ride in the launch that finalises a BatchNorm backward (ttk_bc_bn_bwd_finalize_fold) instead of their own launch:
# 0 none, 1 the depthwise rows, 2 + the slice tiles of the fused early layers, 3 + those of the wide layers (tools/exp/ab_fold_finalize.py)
_FOLD_WITH_FINALIZE = 2


def part_buffer(B, H, W, device, blocks, blur=False):
    """One scratch buffer large enough for every layer's [rows][2][C] partial sums."""
    L = _hip.lib()
    rows_pw, rows_dw = L.cdll.ttk_bc_partial_rows_pw, L.cdll.ttk_bc_partial_rows_dw
    h = (H + 1) // 2
    need = L.partial_rows_elementwise(B * h * h * 8) * 2 * 32
    for _, cin, cout, stride in blocks:
        ho = (h - 1) // stride + 1
        need = max(need, rows_dw(B, h, h, cin, stride, 1) * 2 * cin, rows_dw(B, h, h, cin, stride, 0) * 2 * cin)
        if blur and stride == 2:
            need = max(need, rows_dw(B, ho, ho, cin, 1, 0) * 2 * cin, rows_dw(B, ho, ho, cin, 1, 1) * 2 * cin)
        need = max(need, rows_pw(B * ho * ho, cin, cout) * 2 * cout, rows_pw(B * ho * ho, cout, cin) * 2 * cin,
                   L.cdll.ttk_bc_pw_bwd_fused_rows(B * ho * ho, cin, cout) * 2 * cin)
        need = max(need, L.cdll.ttk_bc_partial_rows_pool(B, ho * ho, cout) * 2 * cout)
        h = ho
    return torch.empty(need, dtype=torch.float32, device=device)


def forward_impl(MB, x, params, buffers, momentum, eps, training, frozen=False, blur=None):
    """`MB`: the mobilenet_v1 module (block table, `_Stage`, `_Ctx`, `_BnArena`, `_identity_bn`)."""
    L = _hip.lib()
    blocks = MB._BLOCKS
    blur = list(blur) if blur is not None else [None] * len(blocks)
    _hip.check_tensors([x], "input")  # (what came from outside is checked once; everything else below is allocated here)
    _hip.check_tensors(params, "parameter")
    _hip.check_tensors(buffers, "BatchNorm buffer")
    _hip.check_tensors(blur, "blur kernel")
    p = _hip.fast_ptr
    dev = x.device
    B, _, H, W = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    part = part_buffer(B, H, W, dev, blocks, any(b is not None for b in blur))
    ctx = MB._Ctx()
    ctx.x, ctx.part, ctx.B = x, part, B
    ctx.blur = []
    ctx.frozen = frozen
    ctx.stages, ctx.a_in, ctx.dims = [], [], []
    ctx.bf, ctx.gdt = "bc", _DT
    bns = MB._BnArena([32] + [c for _, cin, cout, _ in blocks for c in (cin, cout)], dev)

    def pivot(bi):
        return p(buffers[3 * bi])

    def finalize(bn, rows, C, count, gamma, beta, bi):
        rm, rv, nbt = buffers[3 * bi], buffers[3 * bi + 1], buffers[3 * bi + 2]
        if training:
            L.call("ttk_bn_fwd_finalize", p(part), pivot(bi), rows, C, count, p(gamma), p(beta), p(rm), p(rv), p(nbt), float(momentum), float(eps), p(bn))
        else:
            L.call("ttk_bn_eval_prepare", p(gamma), p(beta), p(rm), p(rv), float(eps), C, p(bn))

    part_arg = p(part)
    w_pws = [params[3 + 6 * k + 3] for k in range(len(blocks))]
    sizes = [L.cdll.ttk_bc_prepared_bytes(cin, cout) for _, cin, cout, _ in blocks]
    pool = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    ctx.prep = list(torch.split(pool, sizes))
    L.bc_prepare_weights(w_pws, ctx.prep)
    # ---- stem (reference :122-126,161-163)
    y0 = torch.empty((B, Ho, Wo, 32), dtype=_DT, device=dev)
    L.call("ttk_stem_fwd", p(x), p(params[0]), p(y0), part_arg, pivot(0), B, H, W, _BF)
    bn = bns.take(32)
    finalize(bn, L.partial_rows_elementwise(B * Ho * Wo * 8), 32, B * Ho * Wo, params[1], params[2], 0)
    prev = MB._Stage(y0, bn, None)
    ctx.stages.append(prev)
    h, w_ = Ho, Wo
    pi, bi = 3, 1
    rows_pw, rows_dw = L.cdll.ttk_bc_partial_rows_pw, L.cdll.ttk_bc_partial_rows_dw
    for name, cin, cout, stride in blocks:
        w_dw, g_dw, b_dw, w_pw, g_pw, b_pw = params[pi:pi + 6]
        pi += 6
        has_skip = stride == 1 and cin == cout
        ho, wo = (h - 1) // stride + 1, (w_ - 1) // stride + 1
        a_in = torch.empty_like(prev.y) if has_skip else None
        ydw = torch.empty((B, ho, wo, cin), dtype=_DT, device=dev)
        k = len(ctx.dims)
        if blur[k] is not None:
            t = torch.empty((B, ho, wo, cin), dtype=_DT, device=dev)
            L.call("ttk_bc_dw_fwd", p(prev.y), p(prev.bn), p(prev.skip), None, p(blur[k]), p(t), part_arg, None, B, h, w_, cin, stride)
            st_t = MB._Stage(t, MB._identity_bn(cin, dev), None)
            L.call("ttk_bc_dw_fwd", p(t), p(st_t.bn), None, None, p(w_dw), p(ydw), part_arg, pivot(bi), B, ho, wo, cin, 1)
            dw_rows = rows_dw(B, ho, wo, cin, 1, 0)
            ctx.blur.append((st_t, blur[k]))
        else:
            L.call("ttk_bc_dw_fwd", p(prev.y), p(prev.bn), p(prev.skip), p(a_in), p(w_dw), p(ydw), part_arg, pivot(bi), B, h, w_, cin, stride)
            dw_rows = rows_dw(B, h, w_, cin, stride, 0)
            ctx.blur.append(None)
        bn_dw = bns.take(cin)
        finalize(bn_dw, dw_rows, cin, B * ho * wo, g_dw, b_dw, bi)
        ypw = torch.empty((B, ho, wo, cout), dtype=_DT, device=dev)
        M = B * ho * wo
        L.call("ttk_bc_pw_fwd", p(ydw), p(bn_dw), p(ctx.prep[k]), p(ypw), part_arg, pivot(bi + 1), M, cin, cout)
        bn_pw = bns.take(cout)
        finalize(bn_pw, rows_pw(M, cin, cout), cout, M, g_pw, b_pw, bi + 1)
        bi += 2
        ctx.stages.append(MB._Stage(ydw, bn_dw, None))
        prev = MB._Stage(ypw, bn_pw, a_in)
        ctx.stages.append(prev)
        ctx.a_in.append(a_in)
        ctx.dims.append((h, w_, ho, wo, cin, cout, stride, has_skip))
        h, w_ = ho, wo
    C = prev.y.shape[-1]
    feat = torch.empty((B, C), dtype=torch.float32, device=dev)
    L.call("ttk_bc_avgpool_fwd", p(prev.y), p(prev.bn), p(prev.skip), p(feat), B, h * w_, C)
    ctx.HW = h * w_
    return feat, ctx


def backward_impl(MB, ctx, gfeat, params):
    L = _hip.lib()
    _hip.check_tensors([gfeat], "gradient")
    _hip.check_tensors(params, "parameter")
    p = _hip.fast_ptr
    B, part = ctx.B, ctx.part
    blocks = MB._BLOCKS
    offs, total = [], 0
    for q in params:
        offs.append(total)
        total += (q.numel() + 63) // 64 * 64
    offs.append(total)
    arena = torch.zeros(total, dtype=torch.float32, device=gfeat.device)
    grads = [arena[o:o + q.numel()].view(q.shape) for o, q in zip(offs, params)]

    def announce(first, last):
        MB.grad_ready_hook(arena, [(params[i], offs[i], offs[i + 1]) for i in range(first, last)])

    last = ctx.stages[-1]
    C = last.y.shape[-1]

    def bwd_finalize(stage, rows, count, gi):
        Cc = stage.y.shape[-1]
        if ctx.frozen:
            L.call("ttk_bn_bwd_frozen", p(stage.bn), Cc)
            return
        L.call("ttk_bn_bwd_finalize", p(part), rows, Cc, count, p(params[gi]), p(stage.bn), p(grads[gi]), p(grads[gi + 1]), 0)

    rows_pw, rows_dw = L.cdll.ttk_bc_partial_rows_pw, L.cdll.ttk_bc_partial_rows_dw
    # scratch of the pointwise weight gradients (slice partials, folded in a fixed order) and - deterministic mode - of the depthwise / stem ones
    need = max(max(L.cdll.ttk_bc_pw_wgrad_scratch_bytes(B * d[2] * d[3], d[4], d[5]), L.cdll.ttk_bc_pw_bwd_fused_scratch_bytes(B * d[2] * d[3], d[4], d[5]))
               for d in ctx.dims)
    det = MB._DETERMINISTIC
    # the fused depthwise weight gradient ALWAYS goes through workgroup rows + a fixed-order fold (float atomics of hundreds of workgroups on
    # the same 9 x 64 addresses serialise at the memory side: 15 - 110 us per launch measured on an otherwise empty kernel)
    need = max(need, max(rows_dw(B, d[0], d[1], d[4], d[6], 1) * 9 * d[4] * 4 for d in ctx.dims))
    need = max([need] + [rows_dw(B, d[2], d[3], d[4], 1, 1) * 9 * d[4] * 4 for d, bl in zip(ctx.dims, ctx.blur) if bl is not None])
    if det:
        need = max(need, L.cdll.ttk_stem_wgrad_partial_bytes())
    scratch = torch.empty(need // 4, dtype=torch.float32, device=gfeat.device)
    dw_scratch = p(scratch)
    stem_scratch = p(scratch) if det else None

    g = torch.empty(last.y.shape, dtype=_DT, device=last.y.device)
    L.call("ttk_bc_avgpool_bwd", p(gfeat), p(last.y), p(last.bn), p(last.skip), p(g), p(part), B, ctx.HW, C)
    bwd_finalize(last, L.cdll.ttk_bc_partial_rows_pool(B, ctx.HW, C), B * ctx.HW, len(params) - 2)

    for k in range(len(blocks) - 1, -1, -1):
        h, w_, ho, wo, cin, cout, stride, has_skip = ctx.dims[k]
        pi = 3 + 6 * k
        w_dw = params[pi]
        st_prev, st_dw, st_pw = ctx.stages[2 * k], ctx.stages[2 * k + 1], ctx.stages[2 * k + 2]
        a_in = ctx.a_in[k]
        M = B * ho * wo
        # -- pointwise: weight gradient, data gradient (+ bn_dw backward sums)
        g_dw = torch.empty(st_dw.y.shape, dtype=_DT, device=st_dw.y.device)
        fused_rows = L.cdll.ttk_bc_pw_bwd_fused_rows(M, cin, cout) if MB._FUSED_PW_BWD else 0
        defer = (not ctx.frozen) and _FOLD_WITH_FINALIZE >= (2 if fused_rows > 0 else 3)  # the slice tiles are folded by the launch that finalises bn_dw's backward
        dWp = None if defer else p(grads[pi + 3])
        if fused_rows > 0:  # the early layers (HBM-bound): one kernel reads g, y, ydw once for both gradients
            L.call("ttk_bc_pw_bwd_fused", p(g), p(st_pw.y), p(st_pw.bn), p(ctx.prep[k]), p(st_dw.y), p(st_dw.bn), p(g_dw), dWp, p(scratch),
                   p(part), M, cin, cout)
            rows, slices = fused_rows, fused_rows
        else:
            L.call("ttk_bc_pw_bwd_weight", p(g), p(st_pw.y), p(st_pw.bn), p(st_dw.y), p(st_dw.bn), dWp, p(scratch), M, cin, cout)
            L.call("ttk_bc_pw_bwd_data", p(g), p(st_pw.y), p(st_pw.bn), p(ctx.prep[k]), p(st_dw.y), p(st_dw.bn), p(g_dw), p(part), M, cin, cout)
            rows, slices = rows_pw(M, cout, cin), L.cdll.ttk_bc_pw_wgrad_slices(M, cin, cout)
        if defer:
            L.call("ttk_bc_bn_bwd_finalize_fold", p(part), rows, cin, M, p(params[pi + 1]), p(st_dw.bn), p(grads[pi + 1]), p(grads[pi + 2]), 0,
                   p(scratch), slices, cin * cout, p(grads[pi + 3]), 1)
        else:
            bwd_finalize(st_dw, rows, M, pi + 1)
        # -- depthwise: data gradient (+ residual gradient, + producer's bn sums) with the fused weight gradient
        dWd = grads[pi]
        g_prev = torch.empty(st_prev.y.shape, dtype=_DT, device=st_prev.y.device)
        if ctx.blur[k] is not None:
            st_t, w_blur = ctx.blur[k]
            g_t = torch.empty(st_t.y.shape, dtype=_DT, device=st_t.y.device)
            L.call("ttk_bc_dw_bwd_data", p(g_dw), p(st_dw.y), p(st_dw.bn), p(w_dw), None, p(st_t.y), p(st_t.bn), None, None, p(g_t), p(part), p(dWd), 1,
                   dw_scratch, B, ho, wo, cin, 1)
            L.call("ttk_bn_bwd_frozen", p(st_t.bn), cin)
            L.call("ttk_bc_dw_bwd_data", p(g_t), p(st_t.y), p(st_t.bn), p(w_blur), None, p(st_prev.y), p(st_prev.bn), p(st_prev.skip), None, p(g_prev),
                   p(part), None, 0, None, B, h, w_, cin, stride)
            bwd_finalize(st_prev, rows_dw(B, h, w_, cin, stride, 1), B * h * w_, pi - 2 if k > 0 else 1)
        else:
            rows = rows_dw(B, h, w_, cin, stride, 1)
            gi = pi - 2 if k > 0 else 1
            if ctx.frozen or _FOLD_WITH_FINALIZE < 1:
                L.call("ttk_bc_dw_bwd_data", p(g_dw), p(st_dw.y), p(st_dw.bn), p(w_dw), p(g) if has_skip else None, p(st_prev.y), p(st_prev.bn),
                       p(st_prev.skip), p(a_in), p(g_prev), p(part), p(dWd), 1, dw_scratch, B, h, w_, cin, stride)
                bwd_finalize(st_prev, rows, B * h * w_, gi)
            else:
                # the kernel leaves its weight-gradient rows unfolded (dw_accumulate = 2); ONE launch finalises the producer's BatchNorm backward
                # AND folds them (two dependent few-microsecond launches less per layer)
                L.call("ttk_bc_dw_bwd_data", p(g_dw), p(st_dw.y), p(st_dw.bn), p(w_dw), p(g) if has_skip else None, p(st_prev.y), p(st_prev.bn),
                       p(st_prev.skip), p(a_in), p(g_prev), p(part), p(dWd), 2, dw_scratch, B, h, w_, cin, stride)
                L.call("ttk_bc_bn_bwd_finalize_fold", p(part), rows, cin, B * h * w_, p(params[gi]), p(st_prev.bn), p(grads[gi]), p(grads[gi + 1]), 0,
                       dw_scratch, rows, 9 * cin, p(dWd), 1)
        g = g_prev
        if MB.grad_ready_hook is not None:
            announce(pi, pi + 6)
    st0 = ctx.stages[0]
    _, _, H, W = ctx.x.shape
    L.call("ttk_stem_bwd_weight", p(g), p(st0.y), p(st0.bn), p(ctx.x), p(grads[0]), 1, stem_scratch, B, H, W, _BF)
    if MB.grad_ready_hook is not None:
        announce(0, 3)
    return grads
