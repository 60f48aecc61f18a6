"""torch.autograd.Function wrappers around the heads / loss entry points of libttk_hip.so.

Every function here requires CUDA tensors and the HIP library: there is no PyTorch fallback (the
module API in losses.py / negloglikelihood.py / models.py raises for CPU tensors in training)."""
from __future__ import annotations

import os
import threading

import torch
from torch.autograd import Function

from .. import _hip

# ---------------------------------------------------------------------------------------------
# launch batching of the loss ops (ttk_loss_batch)
# ---------------------------------------------------------------------------------------------
class _Batch:
    """Loss launches collected instead of issued; flush() issues them as one ttk_loss_batch launch.  Only ops that do not
    depend on one another may meet in a batch (train.default_compute_loss: the per-sample values of all criterions in the
    forward pass, their gradients in the backward pass)."""

    def __init__(self):
        self.ops, self.keep, self.records = [], [], []
        self.deferring = False  # set while a Function driven by apply() / BatchedLossFn runs: only ITS launches are deferred

    def flush(self):
        ops, self.ops = self.ops, []
        if ops:
            _hip.lib().loss_batch(ops)
        self.keep = []  # the launch is on the stream: the caching allocator may recycle the temporaries behind it


_BATCHING = True  # False (set by tools / tests): one launch per loss op, the A/B of the batched launch
_TLS = threading.local()  # the open batch of this thread (backward runs on autograd's worker thread)


def _batch() -> _Batch | None:
    return getattr(_TLS, "batch", None)


class loss_batch:
    """with loss_batch() as b: loss Functions applied through `apply` below defer their launches into b."""

    def __enter__(self):
        self.prev, _TLS.batch = _batch(), _Batch()
        return _TLS.batch

    def __exit__(self, *exc):
        _TLS.batch = self.prev
        return False


class unbatched:
    """with unbatched(): loss Functions launch one by one again (inside or outside a loss_batch block)."""

    def __enter__(self):
        self.prev, _TLS.batch = _batch(), None

    def __exit__(self, *exc):
        _TLS.batch = self.prev
        return False


def _p(t):
    b = _batch()
    if b is not None and b.deferring and t is not None:
        b.keep.append(t)  # a deferred launch reads this pointer later: the tensor must outlive the flush
    return _hip.ptr(t)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("this loss/head runs in HIP kernels on the MI355X: CUDA tensors required (no CPU fallback)")
    return t.detach().to(torch.float32).contiguous()


def _call(name, *args):
    b = _batch()
    if b is not None and b.deferring and name in _hip.LOSS_BATCH_OPS:
        b.ops.append((name, args))
        return
    _hip.lib().call(name, *args)


class _Ctx:
    """What a loss Function's forward / backward use of the autograd context, for Functions driven by BatchedLossFn."""

    def __init__(self):
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, _flag):
        pass


def apply(fn, *args):
    """fn.apply(*args) - or, inside loss_batch(), the same forward with its launch deferred: the returned per-sample values are
    filled in by the batch's flush and carry no autograd history; the batch records what BatchedLossFn needs to run
    fn.backward.  Only for call sites that hand the result back untouched (nothing may read it before the flush)."""
    b = _batch() if _BATCHING else None
    if b is None or not all(a.is_cuda for a in args if isinstance(a, torch.Tensor)):
        return fn.apply(*args)
    ctx = _Ctx()
    b.deferring = True
    try:
        with torch.no_grad():
            v = fn.forward(ctx, *args)
    finally:
        b.deferring = False
    b.records.append((fn, ctx, args, v))
    return v


class BatchedLossFn(Function):
    """scale * sum_k w_k * sum_i sample_w_k[i] * val_k[i] over all loss terms of a step, with the terms' own forward and backward
    kernels batched: forward = one ttk_loss_batch launch (the deferred ops of `batch`) + one weighted sum; backward = one
    launch for d loss / d values + one ttk_loss_batch launch for every term's gradient.  `records[k]` (fn, ctx, args, v)
    describes deferred term k, inputs `tensors[slots[k][j]]` are its differentiable arguments; the trailing `ordinary` values are
    terms computed the usual way (autograd history of their own)."""

    @staticmethod
    def forward(ctx, batch, records, slots, scalars, sample_ws, scale, n_inputs, *tensors):
        batch.flush()
        ordinary = list(tensors[n_inputs:])
        vals = [r[3] for r in records] + ordinary
        loss = WeightedSumFn.forward(ctx, scalars, sample_ws, scale, *vals)
        ctx.rec = (records, slots, n_inputs, len(ordinary))
        return loss

    @staticmethod
    def backward(ctx, g):
        records, slots, n_inputs, n_ord = ctx.rec
        gvals = WeightedSumFn.backward(ctx, g)[3:]
        grads = [None] * n_inputs
        with loss_batch() as b:
            b.deferring = True
            outs = [fn.backward(fctx, gv) for (fn, fctx, _, _), gv in zip(records, gvals)]
            b.deferring = False
            b.flush()
        for sl, out in zip(slots, outs):
            out = out if isinstance(out, tuple) else (out,)
            for pos, j in sl:  # argument position of the Function -> index into the distinct inputs
                if out[pos] is not None:
                    grads[j] = out[pos] if grads[j] is None else grads[j] + out[pos]
        return (None,) * 7 + tuple(grads) + tuple(gvals[len(records):])


# Data-parallel hook (trackertraincode.parallel.install): called as hook(arena, [(param, lo, hi), ...]) when the fused
# heads' parameter gradients - slices of one flat arena - are final (before the backbone's backward starts).
grad_ready_hook = None


# ---------------------------------------------------------------------------------------------
# heads
# ---------------------------------------------------------------------------------------------
class HeadsFn(Function):
    """feat[B,F] -> (roi, coord, rot, unnormalized_quat[, coord_scales, pose_scales_tril][, pt3d_68, shapeparam])"""

    @staticmethod
    def forward(ctx, feat, ids, unc, pt, use_offset, rot6d, keypts, keyeig, P, Pk, *lin):
        feat = _f32c(feat)
        B, F = feat.shape
        ws, bs = [_f32c(w) for w in lin[0::2]], [_f32c(b) for b in lin[1::2]]
        dev = feat.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        rows = [w.shape[0] for w in ws]
        NZ = sum(rows)
        wcat, bcat = new(NZ, F), new(NZ)  # all heads' weights / biases stacked, one launch
        _hip.lib().multi_copy(ws + bs, list(torch.split(wcat, rows)) + list(torch.split(bcat, rows)))
        z, roi, coord = new(B, NZ), new(B, 4), new(B, 3)
        rot, qu = (new(B, 3, 3), new(B, 6)) if rot6d else (new(B, 4), new(B, 4))  # 6D head: matrices + raw 6D features
        Lc, Lr = (new(B, 3, 3), new(B, 3, 3)) if unc else (None, None)
        pts, shp = (new(B, 68, 3), new(B, 50)) if pt else (None, None)
        ids32 = None if ids is None else ids.to(device=dev, dtype=torch.int32).contiguous()
        Pc = _f32c(P) if (use_offset and P is not None) else None
        Pkc = _f32c(Pk) if (use_offset and pt and Pk is not None) else None
        kp = _f32c(keypts) if pt else None
        ke = _f32c(keyeig) if pt else None
        _call("ttk_heads_fwd", _p(feat), _p(wcat), _p(bcat), _p(ids32), _p(Pc), _p(Pkc), _p(kp), _p(ke), B, F, NZ, int(unc),
              int(pt), int(use_offset), int(rot6d), _p(z), _p(roi), _p(coord), _p(rot), _p(qu), _p(Lc), _p(Lr), _p(pts), _p(shp))
        ctx.save_for_backward(feat, wcat, z, ids32, Pc, Pkc, kp, ke)
        ctx.cfg = (bool(unc), bool(pt), bool(use_offset), bool(rot6d), [w.shape[0] for w in ws])
        ctx.param_refs = (lin, P, Pk)  # the nn.Parameters themselves (identity is kept through apply): for the data-parallel hook
        outs = [roi, coord, rot, qu]
        if unc:
            outs += [Lc, Lr]
        if pt:
            outs += [pts, shp]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        feat, wcat, z, ids32, Pc, Pkc, kp, ke = ctx.saved_tensors
        unc, pt, use_offset, rot6d, rows = ctx.cfg
        B, F = feat.shape
        NZ = wcat.shape[0]
        dev = feat.device
        g = [_f32c(t) for t in gouts]
        g_roi, g_coord, g_rot, g_qu = g[:4]
        k = 4
        g_Lc = g_Lr = g_pts = g_shp = None
        if unc:
            g_Lc, g_Lr = g[k], g[k + 1]
            k += 2
        if pt:
            g_pts, g_shp = g[k], g[k + 1]
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        dz, dprow, dfeat = new(B, NZ), new(B, 8), new(B, F)
        # every parameter gradient of the heads in ONE flat arena (one all-reduce range for data-parallel replicas)
        arena = new(NZ * F + NZ + 64)
        dwcat, dbcat = arena[:NZ * F].view(NZ, F), arena[NZ * F:NZ * F + NZ]
        dP = arena[NZ * F + NZ:NZ * F + NZ + 32].view(8, 4) if Pc is not None else None
        dPk = arena[NZ * F + NZ + 32:NZ * F + NZ + 64].view(8, 4) if Pkc is not None else None
        if (dP is None) != (dPk is None):
            arena[NZ * F + NZ:].zero_()  # one table only: the other slot lies inside the range the data-parallel hook announces (merged lo..hi)
        _call("ttk_heads_bwd", _p(feat), _p(wcat), _p(z), _p(ids32), _p(Pc), _p(Pkc), _p(kp), _p(ke), B, F, NZ, int(unc), int(pt),
              int(use_offset), int(rot6d), _p(g_roi), _p(g_coord), _p(g_rot), _p(g_qu), _p(g_Lc), _p(g_Lr), _p(g_pts), _p(g_shp), _p(dz),
              _p(dprow), _p(dfeat), _p(dwcat), _p(dbcat), _p(dP), _p(dPk))
        lin_grads = []
        r0 = 0
        for r in rows:
            lin_grads += [dwcat[r0:r0 + r], dbcat[r0:r0 + r]]
            r0 += r
        if grad_ready_hook is not None:
            lin, P, Pk = ctx.param_refs
            entries, r0 = [], 0
            for i, r in enumerate(rows):
                entries += [(lin[2 * i], r0 * F, (r0 + r) * F), (lin[2 * i + 1], NZ * F + r0, NZ * F + r0 + r)]
                r0 += r
            if dP is not None:
                entries.append((P, NZ * F + NZ, NZ * F + NZ + 32))
            if dPk is not None:
                entries.append((Pk, NZ * F + NZ + 32, NZ * F + NZ + 64))
            grad_ready_hook(arena, entries)  # the announced range ends with the last table present: every element of it was written
        # inputs: feat, ids, unc, pt, use_offset, rot6d, keypts, keyeig, P, Pk, *lin
        return (dfeat, None, None, None, None, None, None, None, dP, dPk, *lin_grads)


class DiagScaleFn(Function):
    """DiagonalScaleParameter: hidden[n+1] -> scales[n]"""

    @staticmethod
    def forward(ctx, hidden):
        h = _f32c(hidden)
        n = h.numel() - 1
        out = torch.empty(n, dtype=torch.float32, device=h.device)
        _call("ttk_diag_scale_fwd", _p(h), _p(out), n)
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, gout):
        (h,) = ctx.saved_tensors
        gh = torch.empty_like(h)
        _call("ttk_diag_scale_bwd", _p(h), _p(_f32c(gout)), _p(gh), h.numel() - 1)
        return gh


# ---------------------------------------------------------------------------------------------
# losses: each returns the per-sample vector [n]
# ---------------------------------------------------------------------------------------------
def _vec(n, like):
    return torch.empty(n, dtype=torch.float32, device=like.device)


class Rot6dLossFn(Function):
    """Rot6dReprLoss: 0.75 - 0.25 tr(R tomatrix(t)^T); R [n,3,3], t [n,4] target quaternions (no gradient)"""

    @staticmethod
    def forward(ctx, R, t):
        R, t = _f32c(R), _f32c(t)
        v = _vec(R.shape[0], R)
        _call("ttk_loss_rot6d_fwd", _p(R), _p(t), R.shape[0], _p(v))
        ctx.save_for_backward(t)
        return v

    @staticmethod
    def backward(ctx, gv):
        (t,) = ctx.saved_tensors
        gR = torch.empty((t.shape[0], 3, 3), dtype=torch.float32, device=t.device)
        _call("ttk_loss_rot6d_bwd", _p(t), _p(_f32c(gv)), t.shape[0], _p(gR))
        return gR, None


class Ortho6dFn(Function):
    """Rot6dNormalizationSoftConstraint on the raw 6D features [n,6]"""

    @staticmethod
    def forward(ctx, z):
        z = _f32c(z)
        v = _vec(z.shape[0], z)
        _call("ttk_loss_ortho6d_fwd", _p(z), z.shape[0], _p(v))
        ctx.save_for_backward(z)
        return v

    @staticmethod
    def backward(ctx, gv):
        (z,) = ctx.saved_tensors
        gz = torch.empty_like(z)
        _call("ttk_loss_ortho6d_bwd", _p(z), _p(_f32c(gv)), z.shape[0], _p(gz))
        return gz


class MatToQuatFn(Function):
    """torchquaternion.from_matrix: [n,3,3] -> [n,4]"""

    @staticmethod
    def forward(ctx, m):
        m = _f32c(m)
        q = torch.empty((m.shape[0], 4), dtype=torch.float32, device=m.device)
        _call("ttk_mat_to_quat_fwd", _p(m), m.shape[0], _p(q))
        ctx.save_for_backward(m)
        return q

    @staticmethod
    def backward(ctx, gq):
        (m,) = ctx.saved_tensors
        gm = torch.empty_like(m)
        _call("ttk_mat_to_quat_bwd", _p(m), _p(_f32c(gq)), m.shape[0], _p(gm))
        return gm


class RotLossFn(Function):
    @staticmethod
    def forward(ctx, q, t):
        q, t = _f32c(q), _f32c(t)
        v = _vec(q.shape[0], q)
        _call("ttk_loss_rot_fwd", _p(q), _p(t), q.shape[0], _p(v))
        ctx.save_for_backward(q, t)
        return v

    @staticmethod
    def backward(ctx, gv):
        q, t = ctx.saved_tensors
        gq = torch.empty_like(q)
        _call("ttk_loss_rot_bwd", _p(q), _p(t), _p(_f32c(gv)), q.shape[0], _p(gq))
        return gq, None


class QuatRegFn(Function):
    @staticmethod
    def forward(ctx, q):
        q = _f32c(q)
        v = _vec(q.shape[0], q)
        _call("ttk_loss_quatreg_fwd", _p(q), q.shape[0], _p(v))
        ctx.save_for_backward(q)
        return v

    @staticmethod
    def backward(ctx, gv):
        (q,) = ctx.saved_tensors
        gq = torch.empty_like(q)
        _call("ttk_loss_quatreg_bwd", _p(q), _p(_f32c(gv)), q.shape[0], _p(gq))
        return gq


class MseRowsFn(Function):
    """mean over the trailing dimension(s) of (p - t)^2; p, t: [n, D] (or [n] with D = 1)"""

    @staticmethod
    def forward(ctx, p, t):
        shape = p.shape
        p, t = _f32c(p).reshape(shape[0], -1), _f32c(t).reshape(shape[0], -1)
        n, D = p.shape
        v = _vec(n, p)
        _call("ttk_loss_mse_rows_fwd", _p(p), _p(t), n, D, _p(v))
        ctx.save_for_backward(p, t)
        ctx.shape = shape
        return v

    @staticmethod
    def backward(ctx, gv):
        p, t = ctx.saved_tensors
        gp = torch.empty_like(p)
        _call("ttk_loss_mse_rows_bwd", _p(p), _p(t), _p(_f32c(gv)), p.shape[0], p.shape[1], _p(gp))
        return gp.view(ctx.shape), None


class MseColsFn(Function):
    """MseRowsFn of p[:, c0:c0+nc] against t[:, c0:c0+nc] for p, t: [n, Dt], without materialising the column slices
    (and their slice-backward fill/copy/accumulate kernels): the gradient comes back for the whole of p."""

    @staticmethod
    def forward(ctx, p, t, c0, nc):
        p, t = _f32c(p), _f32c(t)
        n, Dt = p.shape
        v = _vec(n, p)
        _call("ttk_loss_mse_cols_fwd", _p(p), _p(t), n, Dt, int(c0), int(nc), _p(v))
        ctx.save_for_backward(p, t)
        ctx.cols = (int(c0), int(nc))
        return v

    @staticmethod
    def backward(ctx, gv):
        p, t = ctx.saved_tensors
        gp = torch.empty_like(p)
        _call("ttk_loss_mse_cols_bwd", _p(p), _p(t), _p(_f32c(gv)), p.shape[0], p.shape[1], *ctx.cols, _p(gp))
        return gp, None, None, None


def mse_cols(p, t, c0, nc):
    """mean((p[..., c0:c0+nc] - t[..., c0:c0+nc])**2, -1) per sample."""
    if p.dim() == 2 and t.shape == p.shape:
        return apply(MseColsFn, p, t, c0, nc)
    sl = slice(c0, c0 + nc) if nc > 1 else c0
    return apply(MseRowsFn, p[..., sl], t[..., sl])


class SplitRowsFn(Function):
    """Row ranges (one per sub-batch) of several prediction tensors as views; backward assembles the gradient of every
    tensor with one launch, where autograd's slice nodes take a fill, a copy and an accumulation per slice."""

    @staticmethod
    def forward(ctx, sizes, *tensors):
        ctx.set_materialize_grads(False)
        ctx.sizes = tuple(int(n) for n in sizes)
        ctx.like = [(t.shape, t.device) for t in tensors]
        outs = []
        for t in tensors:
            off = 0
            for n in ctx.sizes:
                outs.append(t.narrow(0, off, n))
                off += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        k = len(ctx.sizes)
        used = [j for j in range(len(ctx.like)) if any(g is not None for g in gs[j * k:(j + 1) * k])]
        if not used:
            return (None,) * (1 + len(ctx.like))
        dev = ctx.like[used[0]][1]
        numels = [int(torch.Size(ctx.like[j][0]).numel()) for j in used]
        flat = torch.empty(sum(numels), dtype=torch.float32, device=dev)
        fulls = [f.view(ctx.like[j][0]) for f, j in zip(torch.split(flat, numels), used)]
        srcs, dsts = [], []
        for full, j in zip(fulls, used):
            off = 0
            for i, n in enumerate(ctx.sizes):
                g = gs[j * k + i]
                srcs.append(None if g is None else _f32c(g))
                dsts.append(full.narrow(0, off, n))
                off += n
        _hip.lib().multi_copy(srcs, dsts)
        out = [None] * len(ctx.like)
        for full, j in zip(fulls, used):
            out[j] = full
        return (None, *out)


class WeightedSumFn(Function):
    """scale * sum_k w_k * sum_i sample_w_k[i] * val_k[i] (the loss sum of train.default_compute_loss) - one launch
    forward, one backward, instead of a fill/mul per term plus concatenations and a reduction."""

    @staticmethod
    def forward(ctx, scalars, sample_ws, scale, *vals):
        from ctypes import c_float, c_int, c_void_p
        vals = [_f32c(v).reshape(-1) for v in vals]
        sws = [None if s is None else _f32c(s).reshape(-1) for s in sample_ws]
        for v, s in zip(vals, sws):
            if s is not None and s.numel() != v.numel():
                raise RuntimeError("per-sample weights and loss values differ in length")
        dev = vals[0].device
        chunks = [(i, min(i + 32, len(vals))) for i in range(0, len(vals), 32)]
        parts = torch.empty(len(chunks), dtype=torch.float32, device=dev)
        for c, (a, b) in enumerate(chunks):
            n = b - a
            _call("ttk_weighted_sum_fwd", n, (c_void_p * n)(*[_p(v) for v in vals[a:b]]), (c_void_p * n)(*[_p(s) for s in sws[a:b]]),
                  (c_float * n)(*[float(w) for w in scalars[a:b]]), (c_int * n)(*[v.numel() for v in vals[a:b]]), float(scale),
                  _p(parts[c:c + 1]))
        ctx.meta = (list(map(float, scalars)), sws, float(scale), [v.shape for v in vals], chunks)
        return parts[0] if len(chunks) == 1 else parts.sum()

    @staticmethod
    def backward(ctx, g):
        from ctypes import c_float, c_int, c_void_p
        scalars, sws, scale, shapes, chunks = ctx.meta
        g = _f32c(g).reshape(1)
        counts = [int(s.numel()) for s in shapes]
        flat = torch.empty(sum(counts), dtype=torch.float32, device=g.device)
        gvals = list(torch.split(flat, counts))
        for a, b in chunks:
            n = b - a
            _call("ttk_weighted_sum_bwd", n, _p(g), (c_void_p * n)(*[_p(s) for s in sws[a:b]]), (c_float * n)(*scalars[a:b]),
                  (c_int * n)(*counts[a:b]), scale, (c_void_p * n)(*[_p(x) for x in gvals[a:b]]))
        return (None, None, None, *gvals)


class PointsLossFn(Function):
    @staticmethod
    def forward(ctx, p, t, dim, chin, eye):
        p, t = _f32c(p), _f32c(t)
        n = p.shape[0]
        v = _vec(n, p)
        _call("ttk_loss_points_fwd", _p(p), _p(t), n, int(dim), float(chin), float(eye), _p(v))
        ctx.save_for_backward(p, t)
        ctx.cfg = (int(dim), float(chin), float(eye))
        return v

    @staticmethod
    def backward(ctx, gv):
        p, t = ctx.saved_tensors
        gp = torch.empty_like(p)
        _call("ttk_loss_points_bwd", _p(p), _p(t), _p(_f32c(gv)), p.shape[0], *ctx.cfg, _p(gp))
        return gp, None, None, None, None


class NllRotFn(Function):
    @staticmethod
    def forward(ctx, q, t, L):
        q, t, L = _f32c(q), _f32c(t), _f32c(L)
        v = _vec(q.shape[0], q)
        _call("ttk_loss_nllrot_fwd", _p(q), _p(t), _p(L), q.shape[0], _p(v))
        ctx.save_for_backward(q, t, L)
        return v

    @staticmethod
    def backward(ctx, gv):
        q, t, L = ctx.saved_tensors
        gq, gL = torch.empty_like(q), torch.empty_like(L)
        _call("ttk_loss_nllrot_bwd", _p(q), _p(t), _p(L), _p(_f32c(gv)), q.shape[0], _p(gq), _p(gL))
        return gq, None, gL


class NllCoordFn(Function):
    @staticmethod
    def forward(ctx, c, t, L):
        c, t, L = _f32c(c), _f32c(t), _f32c(L)
        v = _vec(c.shape[0], c)
        _call("ttk_loss_nllcoord_fwd", _p(c), _p(t), _p(L), c.shape[0], _p(v))
        ctx.save_for_backward(c, t, L)
        return v

    @staticmethod
    def backward(ctx, gv):
        c, t, L = ctx.saved_tensors
        gc, gL = torch.empty_like(c), torch.empty_like(L)
        _call("ttk_loss_nllcoord_bwd", _p(c), _p(t), _p(L), _p(_f32c(gv)), c.shape[0], _p(gc), _p(gL))
        return gc, None, gL


def _dist_nll_fwd(sym, ctx, mu, sigma, x, points, dim, chin, eye):
    mu, sigma, x = _f32c(mu), _f32c(sigma), _f32c(x)
    n = mu.shape[0]
    per = mu[0].numel()
    v = _vec(n, mu)
    cfg = (n, per, int(points), int(dim), float(chin), float(eye))
    _call(sym + "_fwd", _p(mu), _p(sigma), _p(x), *cfg, _p(v))
    ctx.save_for_backward(mu, sigma, x)
    ctx.cfg = cfg
    return v


def _dist_nll_bwd(sym, ctx, gv):
    mu, sigma, x = ctx.saved_tensors
    gmu, gsg = torch.empty_like(mu), torch.empty_like(sigma)
    _call(sym + "_bwd", _p(mu), _p(sigma), _p(x), _p(_f32c(gv)), *ctx.cfg, _p(gmu), _p(gsg))
    return gmu, gsg, None, None, None, None, None


class NormalNllFn(Function):
    """-mean Normal(mu, sigma).log_prob(x).  points=False: tensors [n, D]; points=True: [n, 68, 3] with the
    first `dim` coordinates and the chin/eye point weights."""

    @staticmethod
    def forward(ctx, mu, sigma, x, points, dim, chin, eye):
        return _dist_nll_fwd("ttk_loss_normal", ctx, mu, sigma, x, points, dim, chin, eye)

    @staticmethod
    def backward(ctx, gv):
        return _dist_nll_bwd("ttk_loss_normal", ctx, gv)


class LaplaceNllFn(Function):
    """-mean Laplace(mu, b).log_prob(x) with NormalNllFn's layouts (distribution="laplace", reference negloglikelihood.py:68-69)."""

    @staticmethod
    def forward(ctx, mu, b, x, points, dim, chin, eye):
        return _dist_nll_fwd("ttk_loss_laplace", ctx, mu, b, x, points, dim, chin, eye)

    @staticmethod
    def backward(ctx, gv):
        return _dist_nll_bwd("ttk_loss_laplace", ctx, gv)


ELEM_KINDS = {"l2": 0, "l1": 1, "smooth_l1": 2}  # TTK_ELEM_*; LOSS_OBJECT_MAP of the reference's losses.py:16-21
SMOOTH_L1_BETA = 0.01                             # torch.nn.SmoothL1Loss(beta=0.01) there


class ElemLossFn(Function):
    """v[s] = sum_d colw[d] * f_kind(p[s, d] - t[s, d]); p, t: [n, ...] flattened to rows, colw: device vector of the row length."""

    @staticmethod
    def forward(ctx, p, t, colw, kind):
        shape = p.shape
        p, t = _f32c(p).reshape(shape[0], -1), _f32c(t).reshape(shape[0], -1)
        n, D = p.shape
        assert colw.numel() == D and colw.is_cuda and colw.dtype == torch.float32
        v = _vec(n, p)
        _call("ttk_loss_elem_fwd", _p(p), _p(t), _p(colw), n, D, int(kind), SMOOTH_L1_BETA, _p(v))
        ctx.save_for_backward(p, t, colw)
        ctx.cfg = (shape, int(kind))
        return v

    @staticmethod
    def backward(ctx, gv):
        p, t, colw = ctx.saved_tensors
        shape, kind = ctx.cfg
        gp = torch.empty_like(p)
        _call("ttk_loss_elem_bwd", _p(p), _p(t), _p(colw), _p(_f32c(gv)), p.shape[0], p.shape[1], kind, SMOOTH_L1_BETA, _p(gp))
        return gp.view(shape), None, None, None


class RotGeodesicFn(Function):
    """smooth_geodesic_distance of the reference's losses.py:24-32."""

    @staticmethod
    def forward(ctx, q, t):
        q, t = _f32c(q), _f32c(t)
        v = _vec(q.shape[0], q)
        _call("ttk_loss_rot_geodesic_fwd", _p(q), _p(t), q.shape[0], _p(v))
        ctx.save_for_backward(q, t)
        return v

    @staticmethod
    def backward(ctx, gv):
        q, t = ctx.saved_tensors
        gq = torch.empty_like(q)
        _call("ttk_loss_rot_geodesic_bwd", _p(q), _p(t), _p(_f32c(gv)), q.shape[0], _p(gq))
        return gq, None


class GmmNllFn(Function):
    @staticmethod
    def forward(ctx, x, ck, mu, sinv, fudge):
        x = _f32c(x)
        n, K = x.shape[0], ck.shape[0]
        v = _vec(n, x)
        post = torch.empty((n, K), dtype=torch.float64, device=x.device)
        _call("ttk_loss_gmm_fwd", _p(x), _p(ck), _p(mu), _p(sinv), K, float(fudge), n, _p(v), _p(post))
        ctx.save_for_backward(x, mu, sinv, post)
        ctx.fudge = float(fudge)
        return v

    @staticmethod
    def backward(ctx, gv):
        x, mu, sinv, post = ctx.saved_tensors
        gx = torch.empty_like(x)
        _call("ttk_loss_gmm_bwd", _p(x), _p(mu), _p(sinv), _p(post), post.shape[1], ctx.fudge, _p(_f32c(gv)), x.shape[0], _p(gx))
        return gx, None, None, None, None
