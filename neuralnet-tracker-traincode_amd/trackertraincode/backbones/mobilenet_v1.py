"""MobileNetV1-style backbone of the pose estimator on MI355X.

Drop-in for the reference's `trackertraincode/backbones/mobilenet_v1.py` (`MobileNet` :95-189,
`DepthWiseBlock` :36-92): same constructor signature, attribute / state-dict names and return
convention `(features[B,F], [z65, z33, z17, z9, z5])`, but the arithmetic is one autograd node whose
forward and backward launch the HIP kernels of ../../csrc through the C-ABI (include/ttk.h):

    stem 5x5 s2 -> 13 x (depthwise 3x3 -> pointwise 1x1 on fp32 MFMA) -> global average pool

Data layout in HBM: activations are fp32 in channel blocks of 32, `[C/32][B*H*W][32]` (include/ttk.h, "Activation layout"): the
tensors below are allocated with the nominal shape `[B, H, W, C]` but only ever travel from kernel to kernel; what PyTorch sees
(the pooled features, the intermediate maps through ttk_bn_act) is plain.  Every conv writes its RAW output once, BatchNorm statistics
come out of the conv's epilogue, and BN + ReLU (+ residual) are applied by the consumer while
loading (forward) - backward mirrors this with three per-channel coefficients (see include/ttk.h).

There is no PyTorch fallback for training: on CUDA tensors the HIP path runs or raises.
CPU tensors are accepted in eval mode only (the ONNX-export / checkpoint-inspection surface of
scripts/export_model.py), through plain torch ops.
"""
from __future__ import annotations

import math
import os
from typing import NamedTuple

import torch
import torch.nn as nn
import torch.nn.functional as F

import sys

from .. import _hip
from ..neuralnets.modelcomponents import BlurPool2D
from . import _mobilenet_bc

_THIS = sys.modules[__name__]

__all__ = ["MobileNet", "DepthWiseBlock"]

NormalizationLayer = nn.BatchNorm2d
ActivationFunc = nn.ReLU

# (name, in, out, stride) of the 13 depthwise-separable blocks - reference mobilenet_v1.py:128-140
_BLOCKS = (
    ("dw2_1", 32, 64, 1), ("dw2_2", 64, 128, 2), ("dw3_1", 128, 128, 1), ("dw3_2", 128, 256, 2),
    ("dw4_1", 256, 256, 1), ("dw4_2", 256, 512, 2), ("dw5_1", 512, 512, 1), ("dw5_2", 512, 512, 1),
    ("dw5_3", 512, 512, 1), ("dw5_4", 512, 512, 1), ("dw5_5", 512, 512, 1), ("dw5_6", 512, 1024, 2),
    ("dw6", 1024, 1024, 1),
)
_INTERMEDIATES = ("dw2_1", "dw3_1", "dw4_1", "dw5_5", "dw6")


class DepthWiseBlock(nn.Module):
    """Parameter container with the reference's names (conv_dw, bn_dw, conv_sep, bn_sep); the
    arithmetic of `forward` lives in the fused backbone function below.  Reference :36-92."""

    def __init__(self, inplanes, planes, stride=1, momentum=0.1, stochastic_depth=None, use_blurpool=True):
        super().__init__()
        assert stride in (1, 2)
        self.inplanes, self.planes = inplanes, planes = int(inplanes), int(planes)
        self.stride = stride
        if stochastic_depth:
            raise NotImplementedError("stochastic_depth is never enabled by the reference's pose estimator")
        self.blurpool = stride == 2 and bool(use_blurpool)
        if self.blurpool:  # reference :43-55: blur + downsample, then the depthwise conv at stride 1 (state-dict names conv_dw.0.kernel, conv_dw.1.weight)
            self.conv_dw = nn.Sequential(
                BlurPool2D(kernel_size=3, stride=2, channels=inplanes),
                nn.Conv2d(inplanes, inplanes, kernel_size=3, padding=1, stride=1, groups=inplanes, bias=False),
            )
        else:
            self.conv_dw = nn.Conv2d(inplanes, inplanes, kernel_size=3, padding=1, stride=stride, groups=inplanes, bias=False)
        self.bn_dw = NormalizationLayer(inplanes, momentum=momentum)
        self.conv_sep = nn.Conv2d(inplanes, planes, kernel_size=1, stride=1, padding=0, bias=False)
        self.bn_sep = NormalizationLayer(planes, momentum=momentum)
        self.relu = nn.ReLU(inplace=True)
        self.skip_connection = not (stride != 1 or inplanes != planes)
        self.stochastic_depth = None

    @property
    def dw_weight(self):
        """The trainable depthwise weight [C,1,3,3] (conv_dw.weight, or conv_dw.1.weight behind a BlurPool2D)."""
        return self.conv_dw[1].weight if self.blurpool else self.conv_dw.weight

    def forward(self, x):  # eval/export path in plain torch ops (CPU); training goes through MobileNet
        out = self.relu(self.bn_dw(self.conv_dw(x)))
        out = self.bn_sep(self.conv_sep(out))
        if self.skip_connection:
            out = out + x
        return self.relu(out)


# Precision of the TRAINING kernels, an attribute of every MobileNet INSTANCE (`MobileNet.precision`, `MobileNet.set_precision`; not part of
# `get_config()` / the state dict, so checkpoints load in the reference unchanged):
#   "fp32"          the reference's precision - activations, gradients and arithmetic fp32 (the default, `bench.py`'s headline);
#   "bf16-compute"  BASELINE config 5's bf16 leg: activation-sized tensors and their gradients bf16 in 64-channel blocks AND bf16 operands of
#                   the pointwise products (one MFMA product, fp32 accumulation); launch sequences in _mobilenet_bc.py, kernels csrc/bc_*.hip.
# The storage-only variants of rounds 2-4 ("bf16" / "bf16-all": bf16 tensors under the fp32 kernels, slower than fp32) are retired.
_PRECISIONS = ("fp32", "bf16-compute")
_RETIRED = ("bf16", "bf16-all")
_DEFAULT_PRECISION = "fp32"


def _check_precision(mode):
    mode = {torch.float32: "fp32", "f32": "fp32"}.get(mode, mode)
    if mode in _RETIRED or mode is torch.bfloat16:
        raise ValueError(f'precision "{mode}" (bf16 storage under the fp32 kernels) was retired: it ran slower than fp32; use "bf16-compute"')
    if mode not in _PRECISIONS:
        raise ValueError(f'precision must be "fp32" or "bf16-compute", got {mode}')
    return mode


def set_activation_dtype(mode):
    """Default precision of MobileNet instances that have not been given one of their own (`MobileNet.set_precision`): "fp32" or
    "bf16-compute".  Kept for the scripts' `--precision` flag; two networks of different precision in one process use `set_precision`."""
    global _DEFAULT_PRECISION
    _DEFAULT_PRECISION = _check_precision(mode)


def bf16_compute() -> bool:
    """Is the module-wide DEFAULT precision bf16-compute (instances may override it)."""
    return _DEFAULT_PRECISION == "bf16-compute"


def get_activation_dtype():
    """(activation, gradient) storage types of the module-wide default precision."""
    t = torch.bfloat16 if bf16_compute() else torch.float32
    return t, t


_BN_ROWS = 8  # TTK_BN_ROWS: scale, beta, mean, rstd, ga, gb, gmean, (pad) - see include/ttk.h


class _BnArena:
    """The BatchNorm constant blocks bn[TTK_BN_ROWS][C] of all layers of one step, cut from ONE zeroed allocation:
    row TTK_BN_AUX (the operand magnitude bounds of the fp16-split GEMMs, include/ttk.h) must start at zero because
    the kernels raise it with atomicMax - one fill launch per step instead of one per layer."""

    def __init__(self, channels, device):
        self._buf = torch.zeros(sum(_BN_ROWS * c for c in channels), dtype=torch.float32, device=device)
        self._off = 0

    def take(self, C) -> torch.Tensor:
        n = _BN_ROWS * C
        blk = self._buf[self._off:self._off + n].view(_BN_ROWS, C)
        self._off += n
        return blk


class _Stage(NamedTuple):
    y: torch.Tensor            # raw conv output [B,H,W,C]
    bn: torch.Tensor           # BatchNorm constant block [8, C]
    skip: torch.Tensor | None  # residual input added before the ReLU of this stage's output


def _part_buffer(B, H, W, device, blur=False):
    """One scratch buffer large enough for every layer's [rows][2][C] partial sums."""
    L = _hip.lib()
    need = 0
    h = (H + 1) // 2
    need = max(need, L.partial_rows_elementwise(B * h * h * 8) * 2 * 32)
    for _, cin, cout, stride in _BLOCKS:
        ho = (h - 1) // stride + 1
        need = max(need, L.partial_rows_dwconv(B, h, h, cin, stride, True) * 2 * cin)     # dw bwd-data
        need = max(need, L.partial_rows_dwconv(B, h, h, cin, stride, False) * 2 * cin)    # dw fwd
        if blur and stride == 2:  # the stride-1 depthwise conv behind the blur
            need = max(need, L.partial_rows_dwconv(B, ho, ho, cin, 1, False) * 2 * cin, L.partial_rows_dwconv(B, ho, ho, cin, 1, True) * 2 * cin)
        need = max(need, L.partial_rows_gemm(B * ho * ho, cin, cout) * 2 * cout, L.partial_rows_gemm(B * ho * ho, cout, cin, True) * 2 * cin)  # pw fwd / bwd-data
        need = max(need, L.partial_rows_elementwise(B * ho * ho * (cout // 4)) * 2 * cout)  # pool bwd
        h = ho
    return torch.empty(need, dtype=torch.float32, device=device)


class _Ctx:
    """Everything one forward pass leaves behind for its backward."""
    __slots__ = ("x", "stages", "a_in", "part", "dims", "pool_skip", "wparams", "HW", "B", "prep", "bf", "gdt", "frozen", "blur")


_IDENTITY_BN: dict = {}


def _identity_bn(C, device):
    """A fresh constant block of the identity map (scale = rstd = 1, beta = mean = 0): the blurred tensor of a BlurPool block has
    no BatchNorm of its own, but every consumer forms its input as relu(bn(y)) on load - relu(t) = t for a blur of
    non-negative activations.  The backward rows (ga, gb, gmean) are written by ttk_bn_bwd_frozen."""
    key = (device.type, device.index, C)
    if key not in _IDENTITY_BN:
        t = torch.zeros((_BN_ROWS, C), dtype=torch.float32, device=device)
        t[0] = 1.0
        t[3] = 1.0
        _IDENTITY_BN[key] = t
    return _IDENTITY_BN[key].clone()


def _forward_impl(x, params, buffers, momentum, eps, training, frozen=False, blur=None):
    """Launches the forward kernels.  `params`: flat list [conv1.w, bn1.w, bn1.b, (dw.w, bn_dw.w,
    bn_dw.b, pw.w, bn_sep.w, bn_sep.b) x 13]; `buffers`: flat list of (running_mean, running_var,
    num_batches_tracked) per BN in the same order; `blur`: per block, the BlurPool2D kernel as a depthwise weight
    [Cin,1,3,3] (use_blurpool, strided blocks) or None."""
    L = _hip.lib()
    blur = list(blur) if blur is not None else [None] * len(_BLOCKS)
    _hip.check_tensors([x], "input")  # what came from outside is checked once; everything else below is allocated here
    _hip.check_tensors(params, "parameter")
    _hip.check_tensors(buffers, "BatchNorm buffer")
    _hip.check_tensors(blur, "blur kernel")
    p = _hip.fast_ptr
    dev = x.device
    B, _, H, W = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    part = _part_buffer(B, H, W, dev, any(b is not None for b in blur))
    ctx = _Ctx()
    ctx.x, ctx.part, ctx.B = x, part, B
    ctx.blur = []
    ctx.frozen = frozen  # backward through eval-mode BatchNorm: the fixed affine map (ttk_bn_bwd_frozen)
    ctx.stages, ctx.a_in, ctx.dims = [], [], []
    act_dtype = torch.float32
    bf = 0  # TTK_STORE_* bits of the C-ABI: fp32 tensors (the bf16 storage variants under these kernels are retired)
    ctx.bf, ctx.gdt = bf, torch.float32
    bns = _BnArena([32] + [c for _, cin, cout, _ in _BLOCKS for c in (cin, cout)], dev)

    def pivot(bi):
        """The statistics pivot of BatchNorm `bi` (include/ttk.h): its running mean - the producer sums y - pivot, the finalisation adds
        it back (and only then updates the running mean)."""
        return p(buffers[3 * bi]) if _hip.bn_pivot() else None

    def finalize(bn, rows, C, count, gamma, beta, bi):
        rm, rv, nbt = buffers[3 * bi], buffers[3 * bi + 1], buffers[3 * bi + 2]
        if training:
            L.call("ttk_bn_fwd_finalize", p(part), pivot(bi), rows, C, count, p(gamma), p(beta), p(rm), p(rv), p(nbt),
                   float(momentum), float(eps), p(bn))
        else:
            L.call("ttk_bn_eval_prepare", p(gamma), p(beta), p(rm), p(rv), float(eps), C, p(bn))
            # eval-mode statistics: the fp16-split GEMMs (forward here; the backward too when the convolutions train with frozen
            # statistics) still scale their operands by a bound of THIS batch's activations, from the producers' partial sums - without
            # it the pieces are unscaled: values above 65 504 overflow, small activations fall into fp16's subnormal range
            L.call("ttk_bn_frozen_bound", p(part), pivot(bi), rows, C, count, p(bn))

    part_arg = p(part)
    # forward and data-gradient weight operands of all 13 pointwise convs, one launch
    w_pws = [params[3 + 6 * k + 3] for k in range(len(_BLOCKS))]
    sizes = [L.pwconv_prepared_bytes(cin, cout) for _, cin, cout, _ in _BLOCKS]
    pool = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
    ctx.prep = list(torch.split(pool, sizes))
    L.pwconv_prepare_weights(w_pws, ctx.prep)
    # ---- stem (reference :122-126,161-163)
    y0 = torch.empty((B, Ho, Wo, 32), dtype=act_dtype, device=dev)
    L.call("ttk_stem_fwd", p(x), p(params[0]), p(y0), part_arg, pivot(0), B, H, W, bf)
    bn = bns.take(32)
    finalize(bn, L.partial_rows_elementwise(B * Ho * Wo * 8), 32, B * Ho * Wo, params[1], params[2], 0)
    prev = _Stage(y0, bn, None)
    ctx.stages.append(prev)
    h, w_ = Ho, Wo
    pi, bi = 3, 1
    for name, cin, cout, stride in _BLOCKS:
        w_dw, g_dw, b_dw, w_pw, g_pw, b_pw = params[pi:pi + 6]
        pi += 6
        has_skip = stride == 1 and cin == cout
        ho, wo = (h - 1) // stride + 1, (w_ - 1) // stride + 1
        a_in = torch.empty_like(prev.y) if has_skip else None
        ydw = torch.empty((B, ho, wo, cin), dtype=act_dtype, device=dev)
        k = len(ctx.dims)
        if blur[k] is not None:
            # reference :43-55 - BlurPool2D (binomial 3x3, stride 2) then the depthwise conv at stride 1: two launches of the
            # same depthwise kernel; the blurred tensor crosses HBM once with the identity constant block (its partial sums
            # are written to the scratch rows and overwritten by the next launch)
            t = torch.empty((B, ho, wo, cin), dtype=act_dtype, device=dev)
            L.call("ttk_dwconv3x3_fwd", p(prev.y), p(prev.bn), p(prev.skip), None, p(blur[k]), p(t), part_arg, None, B, h, w_, cin, stride, bf)
            st_t = _Stage(t, _identity_bn(cin, dev), None)
            L.call("ttk_dwconv3x3_fwd", p(t), p(st_t.bn), None, None, p(w_dw), p(ydw), part_arg, pivot(bi), B, ho, wo, cin, 1, bf)
            dw_rows = L.partial_rows_dwconv(B, ho, wo, cin, 1, False)
            ctx.blur.append((st_t, blur[k]))
        else:
            L.call("ttk_dwconv3x3_fwd", p(prev.y), p(prev.bn), p(prev.skip), p(a_in), p(w_dw), p(ydw), part_arg, pivot(bi), B, h, w_, cin,
                   stride, bf)
            dw_rows = L.partial_rows_dwconv(B, h, w_, cin, stride, False)
            ctx.blur.append(None)
        if _EXP_TENSOR_HOOK is not None:
            _EXP_TENSOR_HOOK("y", k, ydw)
        bn_dw = bns.take(cin)
        finalize(bn_dw, dw_rows, cin, B * ho * wo, g_dw, b_dw, bi)
        ypw = torch.empty((B, ho, wo, cout), dtype=act_dtype, device=dev)
        M = B * ho * wo
        L.call("ttk_pwconv1x1_fwd", p(ydw), p(bn_dw), None, p(ypw), part_arg, pivot(bi + 1), M, cin, cout, p(ctx.prep[len(ctx.dims)]), bf)
        if _EXP_TENSOR_HOOK is not None:
            _EXP_TENSOR_HOOK("y", k, ypw)
        bn_pw = bns.take(cout)
        finalize(bn_pw, L.partial_rows_gemm(M, cin, cout), cout, M, g_pw, b_pw, bi + 1)
        bi += 2
        ctx.stages.append(_Stage(ydw, bn_dw, None))
        prev = _Stage(ypw, bn_pw, a_in)
        ctx.stages.append(prev)
        ctx.a_in.append(a_in)
        ctx.dims.append((h, w_, ho, wo, cin, cout, stride, has_skip))
        h, w_ = ho, wo
    C = prev.y.shape[-1]
    feat = torch.empty((B, C), dtype=torch.float32, device=dev)
    L.call("ttk_avgpool_fwd", p(prev.y), p(prev.bn), p(prev.skip), p(feat), B, h * w_, C, bf)
    ctx.HW = h * w_
    return feat, ctx


# Data-parallel hook: called as hook(arena, [(param, lo, hi), ...]) whenever a block's parameter gradients - elements
# [lo, hi) of the flat gradient arena - are final, in reverse layer order (trackertraincode.parallel.GradAllReduce.on_ready
# all-reduces the ranges in place).
grad_ready_hook = None


# Experiment hooks (module attributes that tools/exp scripts flip; the product never reads the environment for them):
# _USE_WGRAD_STREAM puts the weight-gradient GEMMs on a second stream beside the data-gradient chain.  It was worth 0.3-0.4 ms/step while
# the kernels left the GPU half empty at their tails; with today's kernels the serial order is 0.3 % faster (same box, alternating runs:
# 9.94 vs 9.98 ms; round 6, the wide layers only, the fused early layers kept: 69.1-69.6 k against 69.7-70.3 k crops/s, tools/exp/ab_wgrad_stream.py -
# the finalisation gaps it could fill are smaller than what two kernels that each want every CU cost one another).  _FUSED_PW_BWD = False selects the
# two-kernel backward of the first pointwise layers.
_USE_WGRAD_STREAM = False
# TTK_DETERMINISTIC=1: every weight-gradient reduction runs in a fixed order (slices of M stored to scratch and folded
# by a second kernel instead of fp32 atomics): two runs of a step give bitwise equal gradients.
_DETERMINISTIC = os.environ.get("TTK_DETERMINISTIC", "0") != "0"
_FUSED_PW_BWD = True
# The fused depthwise weight gradient: 0 = float atomics (the product until round 5), 1 = workgroup rows + their own fold launch (slower: 67.3 k against
# 68.4 k crops/s), 2 = rows folded inside the launch that finalises the producer's BatchNorm backward (ttk_bc_bn_bwd_finalize_fold: 68.9 k, same box,
# tools/exp/ab_dw_rows_fp32.py) - the product.  Deterministic mode keeps its own scratch and fold order.
_DW_WGRAD_ROWS = 2
_SIDE_STREAMS: dict = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


# Experiment hook of tools/soak.py (None in the product): called as _EXP_TENSOR_HOOK(kind, block, tensor) right after a kernel has produced
# a gradient ("g": with respect to a conv output before its BatchNorm) or an activation ("y": a raw conv output) of the fp32 path, e.g. to round it
# to the bf16 grid in place - how much of the bf16-compute path's soak gap each tensor class explains (profiles/r06_soak_rounding_ab.txt).
_EXP_TENSOR_HOOK = None


def _backward_impl(ctx: _Ctx, gfeat, params):
    L = _hip.lib()
    _hip.check_tensors([gfeat], "gradient")
    _hip.check_tensors(params, "parameter")
    p = _hip.fast_ptr
    B, part, bf = ctx.B, ctx.part, ctx.bf
    # one zeroed arena for every parameter gradient (the weight-gradient kernels accumulate atomically): one fill launch
    offs, total = [], 0
    for q in params:
        offs.append(total)
        total += (q.numel() + 63) // 64 * 64
    offs.append(total)
    arena = torch.zeros(total, dtype=torch.float32, device=gfeat.device)
    grads = [arena[o:o + q.numel()].view(q.shape) for o, q in zip(offs, params)]

    def announce(first, last):  # parameters first .. last-1 are final: their slice of the arena (padding included) may travel
        grad_ready_hook(arena, [(params[i], offs[i], offs[i + 1]) for i in range(first, last)])
    last = ctx.stages[-1]
    C = last.y.shape[-1]

    def bwd_finalize(stage: _Stage, rows, count, gi):
        """BatchNorm backward constants of `stage` + its dgamma/dbeta -> grads[gi], grads[gi+1]"""
        Cc = stage.y.shape[-1]
        if ctx.frozen:  # frozen statistics (fine-tuning): no sums, no gamma / beta gradient
            L.call("ttk_bn_bwd_frozen", p(stage.bn), Cc)
            return
        L.call("ttk_bn_bwd_finalize", p(part), rows, Cc, count, p(params[gi]), p(stage.bn), p(grads[gi]), p(grads[gi + 1]), 0)

    main = torch.cuda.current_stream(gfeat.device)
    side = _side_stream(gfeat.device) if (_USE_WGRAD_STREAM and not _DETERMINISTIC) else None
    wg_scratch = None
    if _DETERMINISTIC:
        # one scratch buffer for every weight-gradient reduction of this backward (used one after the other on one stream):
        # slices / workgroups store partial results there and a second kernel folds them in a fixed order
        need = max(L.pwconv_wgrad_partial_bytes(B * d[2] * d[3], d[4], d[5]) for d in ctx.dims)
        need = max(need, L.cdll.ttk_stem_wgrad_partial_bytes())
        need = max(need, max(L.cdll.ttk_pwconv1x1_bwd_fused_partial_bytes(B * d[2] * d[3], d[4], d[5]) for d in ctx.dims))
        need = max(need, max(L.partial_rows_dwconv(B, d[0], d[1], d[4], d[6], True) * 9 * d[4] * 4 for d in ctx.dims))
        need = max([need] + [L.partial_rows_dwconv(B, d[2], d[3], d[4], 1, True) * 9 * d[4] * 4 for d, bl in zip(ctx.dims, ctx.blur) if bl is not None])
        wg_scratch = torch.empty(need // 4, dtype=torch.float32, device=gfeat.device)
    # the weight gradient of the wide pointwise layers (Cin, Cout multiples of 256) always reduces slice partials in a fixed order
    # (csrc/pwconv_r.hip: faster than float atomics for 256 x 256 tiles): its scratch, in every mode
    pw_need = max([L.pwconv_wgrad_scratch_bytes(B * d[2] * d[3], d[4], d[5]) for d in ctx.dims] + [0])
    pw_scratch = wg_scratch if (wg_scratch is not None and wg_scratch.numel() * 4 >= pw_need) else (
        torch.empty(pw_need // 4, dtype=torch.float32, device=gfeat.device) if pw_need else None)
    keep = []
    # The fused depthwise weight gradient of the fp32 kernels: workgroup rows (see _DW_WGRAD_ROWS above; deterministic mode: rows + a fixed-order fold of its own).
    # (the BlurPool blocks and frozen fine-tuning keep float atomics unless deterministic: they pass `wg_scratch`, not these rows)
    dw_rows = wg_scratch
    rows_layers = [d for d, bl in zip(ctx.dims, ctx.blur) if bl is None]
    if dw_rows is None and _DW_WGRAD_ROWS and not ctx.frozen and rows_layers:
        need = max(L.partial_rows_dwconv(B, d[0], d[1], d[4], d[6], True) * 9 * d[4] for d in rows_layers)
        dw_rows = torch.empty(need, dtype=torch.float32, device=gfeat.device)

    g = torch.empty(last.y.shape, dtype=ctx.gdt, device=last.y.device)
    L.call("ttk_avgpool_bwd", p(gfeat), p(last.y), p(last.bn), p(last.skip), p(g), p(part), B, ctx.HW, C, bf)
    bwd_finalize(last, L.partial_rows_elementwise(B * ctx.HW * (C // 4)), B * ctx.HW, len(params) - 2)
    if _EXP_TENSOR_HOOK is not None:
        _EXP_TENSOR_HOOK("g", len(_BLOCKS), g)

    for k in range(len(_BLOCKS) - 1, -1, -1):
        h, w_, ho, wo, cin, cout, stride, has_skip = ctx.dims[k]
        pi = 3 + 6 * k
        w_dw, w_pw = params[pi], params[pi + 3]
        st_prev, st_dw, st_pw = ctx.stages[2 * k], ctx.stages[2 * k + 1], ctx.stages[2 * k + 2]
        a_in = ctx.a_in[k]
        M = B * ho * wo
        # -- pointwise: weight gradient, then data gradient (+ bn_dw backward sums).  The weight gradient has no
        # consumer inside backward, so it runs on a second HIP stream next to the data-gradient chain: its
        # tail (too few tiles left for 256 CUs) and the HBM-bound depthwise kernels fill each other's gaps.
        dW = grads[pi + 3]
        g_dw = torch.empty(st_dw.y.shape, dtype=ctx.gdt, device=st_dw.y.device)
        fused_rows = L.cdll.ttk_pwconv1x1_bwd_fused_rows(M, cin, cout) if (_FUSED_PW_BWD and bf == 0) else 0  # (one kernel for both gradients: main stream, with or without the side stream)
        if fused_rows:
            # the first three pointwise layers (HBM-bound, the largest activations): weight and data gradient in ONE kernel - g,
            # the conv output and the depthwise output are read once instead of twice (csrc/pw_bwd_fused.hip)
            L.call("ttk_pwconv1x1_bwd_fused", p(g), p(st_pw.y), p(st_pw.bn), p(w_pw), p(ctx.prep[k]), p(st_dw.y), p(st_dw.bn), p(g_dw), p(dW),
                   p(wg_scratch), p(part), M, cin, cout)
            bwd_finalize(st_dw, fused_rows, M, pi + 1)
            if _EXP_TENSOR_HOOK is not None:
                _EXP_TENSOR_HOOK("g", k, g_dw)
        elif side is not None:
            ev = torch.cuda.Event()
            ev.record(main)  # g, bn_pw backward constants and the zeroed dW are ready
            side.wait_event(ev)
            with torch.cuda.stream(side):
                L.call("ttk_pwconv1x1_bwd_weight", p(g), p(st_pw.y), p(st_pw.bn), p(st_dw.y), p(st_dw.bn), p(dW),
                       p(pw_scratch) if L.pwconv_wgrad_scratch_bytes(M, cin, cout) else None, M, cin, cout, bf)
                if grad_ready_hook is not None:
                    done = torch.cuda.Event()
                    done.record(side)
            keep.append(g)  # main must not recycle g's memory while the side stream still reads it
        else:
            L.call("ttk_pwconv1x1_bwd_weight", p(g), p(st_pw.y), p(st_pw.bn), p(st_dw.y), p(st_dw.bn), p(dW),
                   p(wg_scratch) if wg_scratch is not None else (p(pw_scratch) if L.pwconv_wgrad_scratch_bytes(M, cin, cout) else None), M, cin, cout, bf)
        if not fused_rows:
            L.call("ttk_pwconv1x1_bwd_data", p(g), p(st_pw.y), p(st_pw.bn), None, p(st_dw.y), p(st_dw.bn), p(g_dw), p(part), M,
                   cin, cout, p(ctx.prep[k]), bf)
            bwd_finalize(st_dw, L.partial_rows_gemm(M, cout, cin, True), M, pi + 1)
            if _EXP_TENSOR_HOOK is not None:
                _EXP_TENSOR_HOOK("g", k, g_dw)
        # -- depthwise: weight gradient, then data gradient (+ residual gradient, + producer's bn sums)
        dWd = grads[pi]  # accumulated by the fused weight-gradient path of bwd_data
        g_prev = torch.empty(st_prev.y.shape, dtype=ctx.gdt, device=st_prev.y.device)
        if ctx.blur[k] is not None:
            # BlurPool block: through the stride-1 conv to the blurred tensor (its "BatchNorm" is the identity: ga = 1, gb = gmean = 0;
            # the mask [t > 0] only drops gradient that the producer's own mask drops as well - t = 0 means every tap was 0), then
            # through the fixed blur kernel (no weight gradient) to the block input
            st_t, w_blur = ctx.blur[k]
            g_t = torch.empty(st_t.y.shape, dtype=ctx.gdt, device=st_t.y.device)
            L.call("ttk_dwconv3x3_bwd_data", p(g_dw), p(st_dw.y), p(st_dw.bn), p(w_dw), None, p(st_t.y), p(st_t.bn), None, None, p(g_t),
                   p(part), p(dWd), 1, p(wg_scratch), B, ho, wo, cin, 1, bf)  # (float atomics unless deterministic: rows + an own fold launch measured slower)
            L.call("ttk_bn_bwd_frozen", p(st_t.bn), cin)
            L.call("ttk_dwconv3x3_bwd_data", p(g_t), p(st_t.y), p(st_t.bn), p(w_blur), None, p(st_prev.y), p(st_prev.bn), p(st_prev.skip),
                   None, p(g_prev), p(part), None, 0, None, B, h, w_, cin, stride, bf)
            bwd_finalize(st_prev, L.partial_rows_dwconv(B, h, w_, cin, stride, True), B * h * w_, pi - 2 if k > 0 else 1)
        elif _DW_WGRAD_ROWS == 2 and not _DETERMINISTIC and not ctx.frozen:
            # workgroup rows, folded by the launch that finalises the producer's BatchNorm backward (one launch instead of atomics + nothing / rows + fold)
            rows, gi = L.partial_rows_dwconv(B, h, w_, cin, stride, True), (pi - 2 if k > 0 else 1)
            L.call("ttk_dwconv3x3_bwd_data", p(g_dw), p(st_dw.y), p(st_dw.bn), p(w_dw), p(g) if has_skip else None, p(st_prev.y),
                   p(st_prev.bn), p(st_prev.skip), p(a_in), p(g_prev), p(part), p(dWd), 2, p(dw_rows), B, h, w_, cin, stride, bf)
            L.call("ttk_bc_bn_bwd_finalize_fold", p(part), rows, cin, B * h * w_, p(params[gi]), p(st_prev.bn), p(grads[gi]), p(grads[gi + 1]), 0,
                   p(dw_rows), rows, 9 * cin, p(dWd), 1)
        else:
            L.call("ttk_dwconv3x3_bwd_data", p(g_dw), p(st_dw.y), p(st_dw.bn), p(w_dw), p(g) if has_skip else None, p(st_prev.y),
                   p(st_prev.bn), p(st_prev.skip), p(a_in), p(g_prev), p(part), p(dWd), 1,
                   p(dw_rows if (_DW_WGRAD_ROWS == 1 and not ctx.frozen) else wg_scratch), B, h, w_, cin, stride, bf)  # (_DW_WGRAD_ROWS = 1: the A/B form, rows + own fold)
            bwd_finalize(st_prev, L.partial_rows_dwconv(B, h, w_, cin, stride, True), B * h * w_, pi - 2 if k > 0 else 1)
        g = g_prev
        if _EXP_TENSOR_HOOK is not None:
            _EXP_TENSOR_HOOK("g", k, g)
        if grad_ready_hook is not None:  # this block's conv + bn_dw gradients and its own bn_sep gradients are final
            if side is not None:
                main.wait_event(done)
            announce(pi, pi + 6)
    st0 = ctx.stages[0]
    _, _, H, W = ctx.x.shape
    L.call("ttk_stem_bwd_weight", p(g), p(st0.y), p(st0.bn), p(ctx.x), p(grads[0]), 1, p(wg_scratch), B, H, W, bf)
    if grad_ready_hook is not None:
        announce(0, 3)
    if side is not None:
        main.wait_stream(side)  # join: everything after backward (clip+Adam) sees the weight gradients
        keep.clear()
    return grads


class _MobileNetFn(torch.autograd.Function):
    """One autograd node for the whole backbone: saves raw conv outputs + BN constants."""

    @staticmethod
    def forward(ctx, x, momentum, eps, buffers, frozen, blur, precision, *params):
        if precision == "bf16-compute":
            feat, c = _mobilenet_bc.forward_impl(_THIS, x, params, buffers, momentum, eps, training=not frozen, frozen=frozen, blur=blur)
        else:
            feat, c = _forward_impl(x, params, buffers, momentum, eps, training=not frozen, frozen=frozen, blur=blur)
        ctx.c = c
        ctx.nparams = len(params)
        ctx.save_for_backward(*params)
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        params = ctx.saved_tensors
        if ctx.c.bf == "bc":
            grads = _mobilenet_bc.backward_impl(_THIS, ctx.c, gfeat.contiguous(), params)
        else:
            grads = _backward_impl(ctx.c, gfeat.contiguous(), params)
        ctx.c = None
        return (None, None, None, None, None, None, None, *grads)


class MobileNet(nn.Module):
    """Reference :95-189.  `forward(x) -> (features, [out1..out5])`."""

    def __init__(self, num_classes=1000, widen_factor=1.0, input_channel=1, momentum=0.1, dropout=0.0,
                 use_blurpool=False, return_only_featuremap=False):
        super().__init__()
        if widen_factor != 1.0 or input_channel != 1:
            raise NotImplementedError("the HIP backbone is built for widen_factor=1.0, input_channel=1 (the pose estimator's configuration, models.py:220)")
        if return_only_featuremap:
            raise NotImplementedError("return_only_featuremap is not used by the pose estimator")

        def block(inplanes, planes, stride=1):
            return DepthWiseBlock(inplanes, planes, stride=stride, momentum=momentum, use_blurpool=use_blurpool)

        self.use_blurpool = bool(use_blurpool)
        self.precision = None  # None = the module-wide default (set_activation_dtype); "fp32" | "bf16-compute" through set_precision
        self.conv1 = nn.Conv2d(input_channel, 32, kernel_size=5, stride=2, padding=2, bias=False)
        self.bn1 = NormalizationLayer(32, momentum=momentum)
        self.relu = ActivationFunc(inplace=True)
        for name, cin, cout, stride in _BLOCKS:
            setattr(self, name, block(cin, cout, stride))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.num_features = 1024
        self.num_intermediate_features = [64, 128, 256, 512, 1024]
        if num_classes:
            self.drop = nn.Dropout(p=dropout) if dropout > 0.0 else nn.Identity()
            self.fc = nn.Linear(1024, num_classes)
        # reference :155-158 - N(0, sqrt(2/(k*k*Cout))) for EVERY conv, 1x1 included
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))

    # ---- precision of the training kernels (an attribute of the instance; not in the checkpoint) ------------------
    def set_precision(self, mode):
        """"fp32" | "bf16-compute" for THIS backbone (None: follow the module-wide default again).  Returns self."""
        self.precision = None if mode is None else _check_precision(mode)
        return self

    def effective_precision(self) -> str:
        return self.precision if getattr(self, "precision", None) is not None else _DEFAULT_PRECISION

    # ---- parameter plumbing -----------------------------------------------------------------
    def _bns(self):
        yield self.bn1
        for name, *_ in _BLOCKS:
            blk = getattr(self, name)
            yield blk.bn_dw
            yield blk.bn_sep

    def _flat_params(self):
        ps = [self.conv1.weight, self.bn1.weight, self.bn1.bias]
        for name, *_ in _BLOCKS:
            b = getattr(self, name)
            ps += [b.dw_weight, b.bn_dw.weight, b.bn_dw.bias, b.conv_sep.weight, b.bn_sep.weight, b.bn_sep.bias]
        return ps

    def _blur_weights(self):
        """Per block: the BlurPool2D kernel as a depthwise weight (use_blurpool, strided blocks) or None."""
        if not self.use_blurpool:
            return None
        return [getattr(self, name).conv_dw[0].depthwise_weight() if getattr(self, name).blurpool else None for name, *_ in _BLOCKS]

    def _flat_buffers(self):
        out = []
        for bn in self._bns():
            out += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
        return out

    def _check(self, x):
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected float32 [B,1,H,W] input, got {tuple(x.shape)} {x.dtype}")
        moms = {bn.momentum for bn in self._bns()}
        epss = {bn.eps for bn in self._bns()}
        if len(moms) != 1 or len(epss) != 1 or None in moms:
            raise NotImplementedError("all BatchNorm layers must share one momentum/eps")
        return moms.pop(), epss.pop()

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        """[B,1,H,W] -> [B,1024]; the path NetworkWithPointHead uses (it discards the intermediates)."""
        if not x.is_cuda:
            return self._forward_torch(x)[0]
        momentum, eps = self._check(x)
        x = x.contiguous()
        bn_training = [bn.training for bn in self._bns()]
        if self.training and all(bn_training):
            return _MobileNetFn.apply(x, momentum, eps, self._flat_buffers(), False, self._blur_weights(), self.effective_precision(), *self._flat_params())
        if any(bn_training):
            raise NotImplementedError("mixed train/eval BatchNorm layers are not supported by the fused backbone")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params()):
            # every BatchNorm in eval mode, convolutions trainable: fine-tuning with frozen statistics (reference
            # models.py:391-394 + modelcomponents.py:208-215, which also freezes gamma and beta)
            if any(p.requires_grad for bn in self._bns() for p in bn.parameters()):
                raise NotImplementedError("eval-mode BatchNorm layers with trainable weight / bias are not built: freeze them "
                                          "(modelcomponents.freeze_norm_stats) or put the layers in training mode")
            return _MobileNetFn.apply(x, momentum, eps, self._flat_buffers(), True, self._blur_weights(), self.effective_precision(), *self._flat_params())
        if self.effective_precision() == "bf16-compute":
            feat, _ = _mobilenet_bc.forward_impl(_THIS, x, [q.detach() for q in self._flat_params()], self._flat_buffers(), momentum, eps, False,
                                                 blur=self._blur_weights())
        else:
            feat, _ = _forward_impl(x, [q.detach() for q in self._flat_params()], self._flat_buffers(), momentum, eps, False, blur=self._blur_weights())
        return feat

    def forward(self, x):
        if not x.is_cuda:
            return self._forward_torch(x)
        feat = self.forward_features(x)
        if hasattr(self, "fc"):
            feat = self.fc(self.drop(feat))
        return feat, self._intermediates(x)

    @torch.no_grad()
    def _intermediates(self, x):
        """[z65, z33, z17, z9, z5] post-activation block outputs (reference :165-186), recomputed without
        autograd on request: the pose network never reads them (models.py:343)."""
        momentum, eps = self._check(x)
        bufs = [b.clone() for b in self._flat_buffers()]  # do not double-update running statistics
        training = self.training
        _, c = _forward_impl(x.contiguous(), [q.detach() for q in self._flat_params()], bufs, momentum, eps, training, blur=self._blur_weights())
        L, p = _hip.lib(), _hip.ptr
        outs = []
        for k, (name, *_r) in enumerate(_BLOCKS):
            if name in _INTERMEDIATES:
                st = c.stages[2 * k + 2]
                a = torch.empty_like(st.y)
                L.call("ttk_bn_act", p(st.y), p(st.bn), p(st.skip), p(a), a.numel() // a.shape[-1], a.shape[-1])
                outs.append(a.permute(0, 3, 1, 2))  # NCHW view of the channels-last copy ttk_bn_act wrote
        return outs

    def _forward_torch(self, x):
        """Plain-torch eval path for CPU tensors (ONNX export / checkpoint inspection).  Training on
        the CPU is not a supported path of this package."""
        if self.training:
            raise RuntimeError("the MI355X training path needs CUDA tensors; CPU tensors are accepted in eval() mode only")
        x = self.relu(self.bn1(self.conv1(x)))
        outs = []
        for name, *_ in _BLOCKS:
            x = getattr(self, name)(x)
            if name in _INTERMEDIATES:
                outs.append(x)
        x = self.avgpool(x)
        x = x.view(x.size(0), -1)
        if hasattr(self, "fc"):
            x = self.fc(self.drop(x))
        return x, outs

    def prepare_finetune(self):
        return [[*ch.parameters()] for ch in self.children()]
