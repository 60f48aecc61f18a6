"""ResNet18 backbone variant on MI355X (reference: trackertraincode/backbones/resnet.py:52-104).

The reference builds `torchvision.models.resnet._resnet(BasicBlock, [2,2,2,2])`, swaps the stem for a 1-channel
`Conv2d(1, 64, 7, 2, 3)`, drops the classifier and appends `Flatten`; torchvision is not vendored there (and absent
here), so `BasicBlock`, the layer plan and the initialisation (kaiming fan_out, `zero_init_residual=True`) are
restated from torchvision.models.resnet.  The module tree keeps torchvision's names, so state dicts interchange:
`layers.0` conv1, `layers.1` bn1, `layers.4..7` the four stages (`<i>.conv1/bn1/conv2/bn2/downsample.{0,1}`).

CUDA tensors in training mode run hand-written HIP kernels through the C-ABI (include/ttk.h): the 7x7 stem, the
max-pool and the residual/BatchNorm elementwise passes in csrc/resnet.hip, every dense 3x3 / strided 1x1 convolution
as an implicit GEMM on the fp16-split producer/consumer kernels (csrc/conv.hip, csrc/pwconv_f16.hip; operand magnitude
bounds travel in row TTK_BN_AUX of the BatchNorm blocks, which therefore come from one zeroed arena per step).  Activations
are channels-last and materialised (post BatchNorm/ReLU); BatchNorm statistics follow the partial-sum -> fp64 finalize
scheme of the MobileNet backbone.  `use_blurpool=True` (reference :31-49,63-66): every BasicBlock's conv1 becomes
Sequential(BlurPool2D(stride), conv3x3(stride 1)) and the max-pool a BlurPool2D(stride 2); the blur runs as its own small kernel on the
channels-last rows (ttk_blur3x3_fwd / _bwd), the convolutions around it are the same implicit GEMMs.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _hip
from ..neuralnets.modelcomponents import BlurPool2D

_BN_ROWS = 8
_PLAN = [(64, 1), (64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1)]  # (planes, stride) per BasicBlock


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock: conv3x3-bn-relu-conv3x3-bn (+ identity | downsample) - relu."""
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None, use_blurpool: bool = False):
        super().__init__()
        self.blurpool = bool(use_blurpool)
        if use_blurpool:  # the reference's CustomBlock (:31-49): blur (+ downsample) first, then the convolution at stride 1 - in EVERY block
            self.conv1 = nn.Sequential(BlurPool2D(kernel_size=3, channels=inplanes, stride=stride), nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False))
        else:
            self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    @property
    def conv1_weight(self):
        return self.conv1[1].weight if self.blurpool else self.conv1.weight

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


_DETERMINISTIC = os.environ.get("TTK_DETERMINISTIC", "0") != "0"  # bitwise reproducible steps (tests/test_determinism_gpu.py)
_LAYOUT_ROWS = 4  # TTK_LAYOUT_ROWS (include/ttk.h): this backbone keeps its activations channels-last, [pixels][C]
_BN_AUX = 7  # TTK_BN_AUX: [0] = TTK_AUX_ACT_BOUND of the activation this BatchNorm forms


def _bn_channels():
    out, cin = [64], 64
    for planes, stride in _PLAN:
        out += [planes, planes] + ([planes] if stride != 1 or cin != planes else [])
        cin = planes
    return out


class _Ctx:
    __slots__ = ("x", "B", "H", "W", "y0", "bn0", "idx", "a1", "blocks", "part", "partd", "last", "frozen", "blur", "a0")


class _Blk:
    """What one BasicBlock leaves behind for backward."""
    __slots__ = ("a_in", "a_bn", "y1", "bn1", "a_mid", "y2", "bn2", "yd", "bnd", "a_out", "h", "ho", "cin", "cout", "stride", "w1b", "w2b", "wdb", "t")


def _part_buffers(B, device):
    L = _hip.lib()
    need = L.partial_rows_elementwise(B * 65 * 65 * 16) * 2 * 64
    h = 33
    for planes, stride in _PLAN:
        ho = (h - 1) // stride + 1
        need = max(need, L.partial_rows_gemm(B * h * h) * 2 * planes, L.partial_rows_elementwise(B * h * h * (planes // 4)) * 2 * planes)
        h = ho
    mk = lambda: torch.empty(need, dtype=torch.float32, device=device)
    return mk(), mk()


def _forward_impl(x, params, buffers, momentum, eps, training=True, blur=False):
    """blur: the use_blurpool variant.  params: [conv1.w, bn1.w, bn1.b, then per block (conv1.w, bn1.w, bn1.b, conv2.w, bn2.w, bn2.b[, ds.w, dsbn.w, dsbn.b])];
    buffers: (running_mean, running_var, num_batches_tracked) per BatchNorm in the same order."""
    L, p = _hip.lib(), _hip.ptr
    dev = x.device
    B, _, H, W = x.shape
    c = _Ctx()
    c.x, c.B, c.H, c.W = x, B, H, W
    c.frozen = not training  # backward through eval-mode BatchNorm: the fixed affine map (ttk_bn_bwd_frozen)
    c.blur, c.a0 = bool(blur), None
    c.part, c.partd = _part_buffers(B, dev)
    part = c.part
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    bi = [0]
    # every BatchNorm block of the step from one zeroed allocation (the kernels raise the bounds of row TTK_BN_AUX with atomicMax)
    from .mobilenet_v1 import _BnArena
    arena = _BnArena(_bn_channels(), dev)
    _bn_work = lambda C, _dev: arena.take(C)
    bound = lambda bn: p(bn[_BN_AUX])  # device float: bound of the activation formed from this BatchNorm

    def pivot():
        """Statistics pivot (include/ttk.h) of the BatchNorm the NEXT finalize() call handles: its running mean, before the update."""
        return p(buffers[3 * bi[0]]) if (training and _hip.bn_pivot()) else None

    def finalize(bn, rows, C, count, gamma, beta, scratch=None):
        rm, rv, nbt = buffers[3 * bi[0]: 3 * bi[0] + 3]
        pv = pivot()
        bi[0] += 1
        if training:
            L.call("ttk_bn_fwd_finalize", p(scratch if scratch is not None else part), pv, rows, C, count, p(gamma), p(beta), p(rm), p(rv),
                   p(nbt), float(momentum), float(eps), p(bn))
        else:  # eval: constants from the running statistics, the partial sums the kernels wrote are ignored
            L.call("ttk_bn_eval_prepare", p(gamma), p(beta), p(rm), p(rv), float(eps), C, p(bn))

    # ---- stem 7x7/s2 + bn + relu + maxpool 3x3/s2 (reference resnet.py:63-66; torchvision ResNet.forward)
    Ho = (H - 1) // 2 + 1
    c.y0 = new(B, Ho, Ho, 64)
    L.call("ttk_stem7_fwd", p(x), p(params[0]), p(c.y0), p(part), pivot(), B, H, W)
    c.bn0 = _bn_work(64, dev)
    finalize(c.bn0, L.partial_rows_elementwise(B * Ho * Ho * 16), 64, B * Ho * Ho, params[1], params[2])
    h = (Ho - 1) // 2 + 1
    c.a1 = new(B, h, h, 64)
    if blur:  # reference :63-66: BlurPool2D(3, channels 64, stride 2) in the max-pool's place, behind relu(bn1(.))
        c.idx = None
        c.a0 = new(B, Ho, Ho, 64)
        L.call("ttk_bn_add_act", p(c.y0), p(c.bn0), None, None, p(c.a0), None, int(not training), B * Ho * Ho, 64)
        L.call("ttk_blur3x3_fwd", p(c.a0), p(c.a1), B, Ho, Ho, 64, 2)
    else:
        c.idx = torch.empty((B, h, h, 64), dtype=torch.uint8, device=dev)
        L.call("ttk_maxpool3x3s2_fwd", p(c.y0), p(c.bn0), p(c.a1), p(c.idx), B, Ho, Ho, 64)

    # forward and data-gradient operands of all 19 convolution weights: one allocation, three launches
    convs, cin, pi = [], 64, 3
    for planes, stride in _PLAN:
        has_ds = stride != 1 or cin != planes
        convs += [params[pi], params[pi + 3]] + ([params[pi + 6]] if has_ds else [])
        pi += 9 if has_ds else 6
        cin = planes
    pool = torch.empty(sum(6 * w.numel() for w in convs), dtype=torch.int16, device=dev)  # 6 bytes per weight and operand (2 fp16 planes + header | 3 bf16 planes), two operands
    wops, off = [], 0
    for w in convs:
        n3 = 3 * w.numel()
        wops.append((pool[off:off + n3], pool[off + n3:off + 2 * n3]))
        off += 2 * n3
    L.conv_prepare_weights(convs, [f for f, _ in wops], [b for _, b in wops])
    wops = iter(wops)

    a_in, a_bn, cin, pi = c.a1, c.bn0, 64, 3
    c.blocks = []
    for bidx, (planes, stride) in enumerate(_PLAN):
        has_ds = stride != 1 or cin != planes
        w1, g1, b1, w2, g2, b2 = params[pi:pi + 6]
        pi += 6
        k = _Blk()
        k.a_in, k.a_bn, k.h, k.cin, k.cout, k.stride = a_in, a_bn, h, cin, planes, stride
        ho = (h - 1) // stride + 1
        k.ho = ho
        M = B * ho * ho
        w1f, k.w1b = next(wops)
        w2f, k.w2b = next(wops)
        wdf, k.wdb = next(wops) if has_ds else (None, None)
        k.y1 = new(B, ho, ho, planes)
        k.t = None
        if blur:  # conv1 = Sequential(BlurPool2D(stride), conv3x3 at stride 1); a blur of a_in is bounded by a_in's bound
            k.t = new(B, ho, ho, cin)
            L.call("ttk_blur3x3_fwd", p(a_in), p(k.t), B, h, h, cin, stride)
            L.call("ttk_conv_fwd", p(k.t), bound(a_bn), p(w1f), p(k.y1), p(part), pivot(), B, ho, ho, cin, planes, 3, 3, 1, 1)
        else:
            L.call("ttk_conv_fwd", p(a_in), bound(a_bn), p(w1f), p(k.y1), p(part), pivot(), B, h, h, cin, planes, 3, 3, stride, 1)
        k.bn1 = _bn_work(planes, dev)
        finalize(k.bn1, L.partial_rows_gemm(M), planes, M, g1, b1)
        k.a_mid = new(B, ho, ho, planes)
        L.call("ttk_bn_add_act", p(k.y1), p(k.bn1), None, None, p(k.a_mid), None, int(not training), M, planes)
        k.y2 = new(B, ho, ho, planes)
        L.call("ttk_conv_fwd", p(k.a_mid), bound(k.bn1), p(w2f), p(k.y2), p(part), pivot(), B, ho, ho, planes, planes, 3, 3, 1, 1)
        k.bn2 = _bn_work(planes, dev)
        finalize(k.bn2, L.partial_rows_gemm(M), planes, M, g2, b2)
        k.yd = k.bnd = None
        if has_ds:
            wd, gd, bd = params[pi:pi + 3]
            pi += 3
            k.yd = new(B, ho, ho, planes)
            L.call("ttk_conv_fwd", p(a_in), bound(a_bn), p(wdf), p(k.yd), p(part), pivot(), B, h, h, cin, planes, 1, 1, stride, 0)
            k.bnd = _bn_work(planes, dev)
            finalize(k.bnd, L.partial_rows_gemm(M), planes, M, gd, bd)
        last = bidx == len(_PLAN) - 1
        if not last:  # the last block's output is only pooled: relu(bn2(y2) + identity) is formed inside the pooling kernel
            k.a_out = new(B, ho, ho, planes)
            L.call("ttk_bn_add_act", p(k.y2), p(k.bn2), p(k.yd if has_ds else a_in), p(k.bnd) if has_ds else None, p(k.a_out),
                   bound(k.bnd if has_ds else a_bn), int(not training), M, planes)
            a_in, a_bn = k.a_out, k.bn2
        else:
            k.a_out = None
        c.blocks.append(k)
        h, cin = ho, planes
    k = c.blocks[-1]
    assert k.yd is None, "the pooled block has an identity shortcut in ResNet18"
    feat = new(B, cin)
    L.call("ttk_avgpool_fwd", p(k.y2), p(k.bn2), p(k.a_in), p(feat), B, h * h, cin, _LAYOUT_ROWS)
    return feat, c


# Data-parallel hook (see mobilenet_v1.grad_ready_hook): hook(arena, [(param, lo, hi), ...]) per finished residual block.
grad_ready_hook = None


def _backward_impl(c: _Ctx, gfeat, params):
    L, p = _hip.lib(), _hip.ptr
    B, part, partd = c.B, c.part, c.partd
    dev = gfeat.device
    new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    bound = lambda bn: p(bn[_BN_AUX])
    # one zeroed arena for every parameter gradient (the weight-gradient kernels accumulate atomically): one fill launch,
    # and contiguous ranges that a data-parallel all-reduce can take in place
    offs, total = [], 0
    for q in params:
        offs.append(total)
        total += (q.numel() + 63) // 64 * 64
    offs.append(total)
    arena = torch.zeros(total, dtype=torch.float32, device=dev)
    grads = [arena[o:o + q.numel()].view(q.shape) for o, q in zip(offs, params)]

    def announce(first, last):
        if grad_ready_hook is not None:
            grad_ready_hook(arena, [(params[i], offs[i], offs[i + 1]) for i in range(first, last)])

    def through_bn(g, y, bn, rows, C):
        """-> (dy | g, None | y pointer): the 3x3 convolutions' two gradients read dy = ga*(g-gmean)+gb*(y-mean) materialised
        once (fp16 kernels: half the operand bytes per pass - the data gradient gathers it nine times)."""
        dy = torch.empty_like(g)
        L.call("ttk_bn_bwd_apply", p(g), p(y), p(bn), p(dy), rows, C)
        return dy, None

    def bwd_finalize(bn, rows, C, count, gi, scratch=None):
        if c.frozen:  # frozen statistics (fine-tuning): no sums, no gamma / beta gradient
            L.call("ttk_bn_bwd_frozen", p(bn), C)
            return
        L.call("ttk_bn_bwd_finalize", p(scratch if scratch is not None else part), rows, C, count, p(params[gi]), p(bn), p(grads[gi]), p(grads[gi + 1]), 0)

    # scratch of the slice-wise (atomic-free, reproducible) weight gradients: one buffer, sized for the largest call
    need = 0
    for k in c.blocks:
        need = max(need, L.conv_wgrad_partial_bytes(B, k.h, k.h, k.cin, k.cout, 3, k.stride), L.conv_wgrad_partial_bytes(B, k.ho, k.ho, k.cout, k.cout, 3, 1),
                   L.conv_wgrad_partial_bytes(B, k.ho, k.ho, k.cin, k.cout, 3, 1) if c.blur else 0,
                   L.conv_wgrad_partial_bytes(B, k.h, k.h, k.cin, k.cout, 1, k.stride) if k.yd is not None else 0)
    wscratch = torch.empty(need // 4, dtype=torch.float32, device=dev) if need else None

    # parameter index of every block's first tensor
    starts, pi, cin = [], 3, 64
    for planes, stride in _PLAN:
        starts.append(pi)
        pi += 9 if (stride != 1 or cin != planes) else 6
        cin = planes

    # ---- average pool backward = gs of the last block (+ its bn2 sums)
    k = c.blocks[-1]
    C, hw = k.cout, k.ho * k.ho
    gs = new(B, k.ho, k.ho, C)
    L.call("ttk_avgpool_bwd", p(gfeat), p(k.y2), p(k.bn2), p(k.a_in), p(gs), p(part), B, hw, C, _LAYOUT_ROWS)
    rows_gs = L.partial_rows_elementwise(B * hw * (C // 4))
    for bidx in range(len(_PLAN) - 1, -1, -1):
        k = c.blocks[bidx]
        pi = starts[bidx]
        has_ds = k.yd is not None
        C, M = k.cout, B * k.ho * k.ho
        # gs = gradient w.r.t. bn2(y2) + shortcut (ReLU mask applied); its partial sums are in part (and partd)
        bwd_finalize(k.bn2, rows_gs, C, M, pi + 4)
        if has_ds:
            bwd_finalize(k.bnd, rows_gs, C, M, pi + 7, scratch=partd)
        # conv2: weight gradient, then data gradient through relu(bn1(y1)) (+ bn1 sums)
        dy2, y2 = through_bn(gs, k.y2, k.bn2, M, C)
        L.call("ttk_conv_bwd_weight", p(dy2), y2, p(k.bn2), p(k.a_mid), bound(k.bn1), p(grads[pi + 3]), p(wscratch), B, k.ho, k.ho, C, C, 3, 3, 1, 1)
        g1 = new(B, k.ho, k.ho, C)
        L.call("ttk_conv_bwd_data", p(dy2), y2, p(k.bn2), p(k.w2b), p(k.y1), p(k.bn1), p(g1), p(part), B, k.ho, k.ho, C, C, 3, 3, 1, 1)
        del dy2
        bwd_finalize(k.bn1, L.partial_rows_gemm(M), C, M, pi + 1)
        # conv1: weight gradient, raw data gradient w.r.t. the block input
        dy1, y1 = through_bn(g1, k.y1, k.bn1, M, C)
        g_in = new(B, k.h, k.h, k.cin)
        if c.blur:  # the convolution saw the blurred input at stride 1; its data gradient goes back through the blur
            L.call("ttk_conv_bwd_weight", p(dy1), y1, p(k.bn1), p(k.t), bound(k.a_bn), p(grads[pi]), p(wscratch), B, k.ho, k.ho, k.cin, C, 3, 3, 1, 1)
            g_t = new(B, k.ho, k.ho, k.cin)
            L.call("ttk_conv_bwd_data", p(dy1), y1, p(k.bn1), p(k.w1b), None, None, p(g_t), None, B, k.ho, k.ho, k.cin, C, 3, 3, 1, 1)
            L.call("ttk_blur3x3_bwd", p(g_t), None, p(g_in), B, k.h, k.h, k.cin, k.stride)
            del g_t
        else:
            L.call("ttk_conv_bwd_weight", p(dy1), y1, p(k.bn1), p(k.a_in), bound(k.a_bn), p(grads[pi]), p(wscratch), B, k.h, k.h, k.cin, C, 3, 3, k.stride, 1)
            L.call("ttk_conv_bwd_data", p(dy1), y1, p(k.bn1), p(k.w1b), None, None, p(g_in), None, B, k.h, k.h, k.cin, C, 3, 3, k.stride, 1)
        del dy1
        if has_ds:
            L.call("ttk_conv_bwd_weight", p(gs), p(k.yd), p(k.bnd), p(k.a_in), bound(k.a_bn), p(grads[pi + 6]), p(wscratch), B, k.h, k.h, k.cin, C, 1, 1, k.stride, 0)
            g_sc = new(B, k.h, k.h, k.cin)
            L.call("ttk_conv_bwd_data", p(gs), p(k.yd), p(k.bnd), p(k.wdb), None, None, p(g_sc), None, B, k.h, k.h, k.cin, C, 1, 1, k.stride, 0)
        else:
            g_sc = gs  # identity shortcut
        if bidx > 0:
            # gradient w.r.t. the previous block's output activation = g_in + g_sc; through its ReLU -> previous gs
            q = c.blocks[bidx - 1]
            Mq = B * q.ho * q.ho
            gs_prev = new(B, q.ho, q.ho, q.cout)
            L.call("ttk_residual_bwd", p(g_in), p(g_sc), p(q.a_out), p(q.y2), p(q.bn2), p(q.yd), p(q.bnd) if q.yd is not None else None,
                   p(gs_prev), p(part), p(partd) if q.yd is not None else None, Mq, q.cout)
            rows_gs = L.partial_rows_elementwise(Mq * (q.cout // 4))
            gs = gs_prev
        else:
            Ho = c.y0.shape[1]
            g0 = new(B, Ho, Ho, 64)
            if c.blur:  # back through the blur (both gradients of a1 summed on load), then through relu(bn1(.)) with its BatchNorm sums
                g_a0 = new(B, Ho, Ho, 64)
                L.call("ttk_blur3x3_bwd", p(g_in), p(g_sc), p(g_a0), B, Ho, Ho, 64, 2)
                L.call("ttk_residual_bwd", p(g_a0), None, p(c.a0), p(c.y0), p(c.bn0), None, None, p(g0), p(part), None, B * Ho * Ho, 64)
            else:
                L.call("ttk_maxpool3x3s2_bwd", p(g_in), p(g_sc), p(c.idx), p(c.y0), p(c.bn0), p(g0), p(part), B, Ho, Ho, 64)
            bwd_finalize(c.bn0, L.partial_rows_elementwise(B * Ho * Ho * 16), 64, B * Ho * Ho, 1)
            sp = None
            if _DETERMINISTIC:  # workgroup partials + a fixed-order fold instead of atomics (the 3x3 convolutions always fold; the
                sp = torch.empty(L.cdll.ttk_stem7_wgrad_partial_bytes(B, c.H, c.W) // 4, dtype=torch.float32, device=dev)  # 1x1 ones under the same flag)
            L.call("ttk_stem7_bwd_weight", p(g0), p(c.y0), p(c.bn0), p(c.x), p(grads[0]), p(sp), B, c.H, c.W)
        # this block's conv / BatchNorm gradients are final (the stem's after the first block)
        announce(pi, pi + (9 if has_ds else 6))
    announce(0, 3)
    return grads


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, momentum, eps, buffers, frozen, blur, *params):
        feat, c = _forward_impl(x, params, buffers, momentum, eps, training=not frozen, blur=blur)
        ctx.c = c
        ctx.save_for_backward(*params)
        return feat

    @staticmethod
    def backward(ctx, gfeat):
        grads = _backward_impl(ctx.c, gfeat.contiguous(), ctx.saved_tensors)
        ctx.c = None
        return (None, None, None, None, None, None, *grads)


class ResNetBackbone(nn.Module):
    """Reference :52-92.  `forward(x) -> (features[B,512], None)`."""

    def __init__(self, use_blurpool: bool = False, zero_init_residual: bool = True):
        super().__init__()
        self.use_blurpool = bool(use_blurpool)
        stages, inplanes = [], 64
        for si in range(4):
            blocks = []
            for bi in range(2):
                planes, stride = _PLAN[2 * si + bi]
                ds = None
                if stride != 1 or inplanes != planes:
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
                blocks.append(BasicBlock(inplanes, planes, stride, ds, use_blurpool=self.use_blurpool))
                inplanes = planes
            stages.append(nn.Sequential(*blocks))
        self.layers = nn.Sequential(
            nn.Conv2d(1, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            BlurPool2D(kernel_size=3, channels=64, stride=2) if self.use_blurpool else nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            *stages, nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())
        self.num_features = 512
        # torchvision ResNet.__init__: kaiming fan_out for every conv, BatchNorm (1, 0), zero-initialised last BatchNorm of
        # each residual branch (reference passes zero_init_residual=True, resnet.py:101)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _bns(self):
        yield self.layers[1]
        for si in range(4, 8):
            for blk in self.layers[si]:
                yield blk.bn1
                yield blk.bn2
                if blk.downsample is not None:
                    yield blk.downsample[1]

    def _flat_params(self):
        ps = [self.layers[0].weight, self.layers[1].weight, self.layers[1].bias]
        for si in range(4, 8):
            for blk in self.layers[si]:
                ps += [blk.conv1_weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias]
                if blk.downsample is not None:
                    ps += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
        return ps

    def _flat_buffers(self):
        out = []
        for bn in self._bns():
            out += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
        return out

    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            if self.training:
                raise RuntimeError("the MI355X training path needs CUDA tensors; CPU tensors are accepted in eval() mode only")
            return self.layers(x)
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 1:
            raise ValueError(f"expected float32 [B,1,H,W] input, got {tuple(x.shape)} {x.dtype}")
        bns = list(self._bns())
        moms, epss = {b.momentum for b in bns}, {b.eps for b in bns}
        if len(moms) != 1 or len(epss) != 1 or None in moms:
            raise NotImplementedError("all BatchNorm layers must share one momentum/eps")
        bn_training = [b.training for b in bns]
        if self.training and all(bn_training):
            return _ResNetFn.apply(x.contiguous(), moms.pop(), epss.pop(), self._flat_buffers(), False, self.use_blurpool, *self._flat_params())
        if any(bn_training):
            raise NotImplementedError("mixed train/eval BatchNorm layers are not supported by the fused backbone")
        if torch.is_grad_enabled() and any(q.requires_grad for q in self._flat_params()):
            # fine-tuning with frozen statistics (reference models.py:391-394 + modelcomponents.py:208-215)
            if any(q.requires_grad for b in bns for q in b.parameters()):
                raise NotImplementedError("eval-mode BatchNorm layers with trainable weight / bias are not built: freeze them "
                                          "(modelcomponents.freeze_norm_stats) or put the layers in training mode")
            return _ResNetFn.apply(x.contiguous(), moms.pop(), epss.pop(), self._flat_buffers(), True, self.use_blurpool, *self._flat_params())
        feat, _ = _forward_impl(x.contiguous(), [q.detach() for q in self._flat_params()], self._flat_buffers(), moms.pop(), epss.pop(),
                                training=False, blur=self.use_blurpool)
        return feat

    def forward(self, x):
        return self.forward_features(x), None


def resnet18(use_blurpool: bool = False):
    """Reference :95-104."""
    return ResNetBackbone(use_blurpool=use_blurpool, zero_init_residual=True)
