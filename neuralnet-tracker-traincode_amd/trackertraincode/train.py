"""Loss plumbing, optimiser, schedules and the training loop (reference: trackertraincode/train.py and
the Lightning pieces of scripts/train_poseestimator.py).

Same public names as the reference for what the hot path uses: LossVal, Criterion, CriterionGroup,
concatenated_lossvals_by_name, default_compute_loss, ExponentialUpThenSteps, LinearUpThenSteps,
SwaCallback.  pytorch-lightning is replaced by `fit()` below, which reproduces the step order the
reference gets from `pl.Trainer(gradient_clip_val=1.0, gradient_clip_algorithm="norm")`:
zero_grad -> forward -> loss -> backward -> global-norm clip -> Adam, LR scheduler stepped per epoch,
SWA update per epoch after `start_epoch`.  The clip+Adam pair is one fused HIP entry point
(`ClipAdam`, csrc/adam.hip).

Difference on purpose: `default_compute_loss` does NOT copy the per-sample loss vectors to the host
and does NOT synchronise the stream every step (reference :433-438); the returned LossVals hold
detached DEVICE tensors (call .cpu() when you want them).
"""
from __future__ import annotations

import ctypes
import itertools
import math
import os
from collections import defaultdict
from typing import Any, Callable, List, NamedTuple, Union

import torch
import torch.nn as nn
from torch import Tensor
from torch.optim.lr_scheduler import LambdaLR

from . import _hip
from .datasets.batch import Batch
from .neuralnets.io import save_model


class LossVal(NamedTuple):
    val: Tensor
    weight: Any
    name: str


class SampleWeight:
    """Per-sample weight of one loss term, `scalar * dataset_weight` (or the scalar broadcast over the sub-batch) -
    what the reference stores as a tensor in `LossVal.weight` after default_compute_loss (:404-412).  The tensor is
    only built when something reads it (`.tensor()`, or arithmetic with a tensor): the training step itself feeds
    (scalar, per_sample) to one fused weighted-sum kernel instead of launching a fill or multiply per term."""

    __slots__ = ("scalar", "per_sample", "_like", "_t")

    def __init__(self, scalar: float, per_sample: Tensor | None, like: Tensor):
        self.scalar, self.per_sample, self._like, self._t = float(scalar), per_sample, like, None

    def tensor(self) -> Tensor:
        if self._t is None:
            like = self._like
            self._t = like.new_full(like.shape, self.scalar) if self.per_sample is None else self.scalar * self.per_sample
        return self._t

    def __mul__(self, other):
        return self.tensor() * other

    __rmul__ = __mul__

    @property
    def shape(self):
        return self._like.shape


def _as_weight_tensor(w):
    return w.tensor() if isinstance(w, SampleWeight) else w


def _concat_groups(groups: dict):
    """{name: [1-D tensors]} -> {name: concatenation}; on the GPU all groups land in one buffer with one launch."""
    flat = [t for ts in groups.values() for t in ts]
    if not flat or not all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 1 and not t.requires_grad for t in flat):
        return {k: torch.concat(ts) for k, ts in groups.items()}
    flat = [t.contiguous() for t in flat]
    buf = torch.empty(sum(t.numel() for t in flat), dtype=torch.float32, device=flat[0].device)
    _hip.lib().multi_copy(flat, list(torch.split(buf, [t.numel() for t in flat])))
    return dict(zip(groups.keys(), torch.split(buf, [sum(t.numel() for t in ts) for ts in groups.values()])))


def concatenated_lossvals_by_name(vals):
    """{name: (values, weights)} concatenated over sub-batches, first-seen order (reference :47-62)."""
    values, weights = defaultdict(list), defaultdict(list)
    for v in vals:
        values[v.name].append(v.val)
        weights[v.name].append(_as_weight_tensor(v.weight))
    values = _concat_groups(values)
    return {k: (values[k], torch.concat(weights[k])) for k in values}


def concatenated_values_by_name(vals):
    """{name: values} only - what the training step logs (no weight tensors are built)."""
    values = defaultdict(list)
    for v in vals:
        values[v.name].append(v.val)
    return _concat_groups(values)


def _split_predictions(preds: dict, sizes):
    """Per sub-batch {key: rows of preds[key]}.  float32 GPU tensors (and the rotation containers around them) are
    split by ONE autograd node whose backward assembles all gradients with a single launch."""
    from .neuralnets import _hipops

    def tensor_of(v):
        t = v if isinstance(v, Tensor) else getattr(v, "value", None)
        ok = isinstance(t, Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() >= 1
        return t if ok else None

    fused = {k: tensor_of(v) for k, v in preds.items()}
    fused = {k: t for k, t in fused.items() if t is not None}
    pieces = _hipops.SplitRowsFn.apply(tuple(sizes), *fused.values()) if fused else ()
    out, offset = [], 0
    for i, n in enumerate(sizes):
        sub = {}
        for k, v in preds.items():
            if k in fused:
                piece = pieces[list(fused).index(k) * len(sizes) + i]
                sub[k] = piece if isinstance(v, Tensor) else type(v)(piece)
            else:
                sub[k] = v[offset:offset + n, ...]
        out.append(sub)
        offset += n
    return out


def _weight_at(w, step):
    return w if isinstance(w, float) else w(step)


class Criterion(NamedTuple):
    name: str
    f: Callable[[Any, Any], Tensor]
    w: Union[float, Callable[[int], float]]

    def evaluate(self, pred, batch, step) -> List[LossVal]:
        return [LossVal(self.f(pred, batch), _weight_at(self.w, step), self.name)]


class CriterionGroup(NamedTuple):
    criterions: List[Union["CriterionGroup", Criterion]]
    name: str = ""
    w: Union[float, Callable[[int], float]] = 1.0

    def evaluate(self, pred, batch, step) -> List[LossVal]:
        w = _weight_at(self.w, step)
        out = []
        for c in self.criterions:
            out += [LossVal(v.val, v.weight * w, self.name + v.name) for v in c.evaluate(pred, batch, step)]
        return out


def default_compute_loss(preds: dict, batch: List[Batch], current_epoch: int, loss):
    """(loss_sum, per-sub-batch lists of LossVal) - reference :372-439.

    Sub-batches are addressed by integer offsets into the concatenated predictions; weights become
    per-sample tensors (scaled by `dataset_weight` when the sub-batch has one); the sum is divided by
    the TOTAL batch size so that a loss a sub-batch does not have counts as zero."""
    from .neuralnets import _hipops

    sizes = [subset.meta.prefixshape[0] for subset in batch]
    split = _split_predictions(preds, sizes)

    def evaluate_all():
        out: list[list[LossVal]] = []
        for subset, subpreds in zip(batch, split):
            crit = loss[subset.meta.tag] if isinstance(loss, dict) else loss
            terms = crit.evaluate(subpreds, subset, current_epoch)
            dw = None
            if "dataset_weight" in subset:
                dw = subset["dataset_weight"]
                assert dw.size(0) == subset.meta.batchsize
            out.append([v._replace(weight=SampleWeight(v.weight, dw, v.val)) for v in terms])
        return out

    # the criterions' HIP kernels are independent of one another: their launches are collected and issued as ONE
    # (neuralnets/_hipops.py: loss_batch / apply / BatchedLossFn), forward here and backward in BatchedLossFn.backward
    with _hipops.loss_batch() as lb:
        all_lossvals = evaluate_all()
    returned = {id(v.val) for terms in all_lossvals for v in terms}
    if any(id(r[3]) not in returned for r in lb.records):
        # A criterion did arithmetic on a deferred per-sample value (scaled it, sliced it, added two losses: the reference's
        # Criterion accepts any callable) - that read memory the batch had not filled yet, and the term would get no gradient.
        # Nothing was launched so far: throw the deferred pass away and evaluate every term with one launch per loss op.
        _warn_once("a Criterion post-processes the value of a batched loss kernel; this step's losses run unbatched "
                   "(wrap the arithmetic into the loss function's autograd to silence this)")
        lb.ops.clear()
        lb.records.clear()
        lb.keep.clear()
        with _hipops.unbatched():
            all_lossvals = evaluate_all()
    batchsize = sum(subset.meta.batchsize for subset in batch)
    flat = list(itertools.chain.from_iterable(all_lossvals))
    if flat and all(v.val.is_cuda and v.val.dtype == torch.float32 for v in flat):
        loss_sum = _batched_loss_sum(lb, flat, 1.0 / batchsize)
    else:  # host-side logic on CPU tensors (tests); the reference's formula
        lb.flush()
        by_name = concatenated_lossvals_by_name(flat)
        loss_sum = torch.concat([v * w for v, w in by_name.values()]).sum() / batchsize
    all_lossvals = [[v._replace(val=v.val.detach()) for v in terms] for terms in all_lossvals]
    return loss_sum, all_lossvals


_WARNED: set = set()


def _warn_once(msg):
    if msg not in _WARNED:
        _WARNED.add(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _batched_loss_sum(lb, flat, scale):
    """The weighted sum over all terms; the terms whose kernels were deferred into `lb` go through BatchedLossFn together."""
    from .neuralnets import _hipops
    by_val = {id(r[3]): r for r in lb.records}
    deferred = [v for v in flat if id(v.val) in by_val]
    ordinary = [v for v in flat if id(v.val) not in by_val]
    if not deferred:
        lb.flush()
        return _hipops.WeightedSumFn.apply([v.weight.scalar for v in flat], [v.weight.per_sample for v in flat], scale, *[v.val for v in flat])
    records = [by_val[id(v.val)] for v in deferred]
    assert len(records) == len(lb.records), "a deferred loss term was dropped"  # (default_compute_loss re-evaluates unbatched before this can happen)
    inputs, index, slots = [], {}, []
    for _fn, _ctx, args, _v in records:  # the distinct differentiable inputs of the deferred terms
        sl = []
        for pos, a in enumerate(args):
            if isinstance(a, Tensor) and a.requires_grad:
                if id(a) not in index:
                    index[id(a)] = len(inputs)
                    inputs.append(a)
                sl.append((pos, index[id(a)]))
        slots.append(sl)
    order = deferred + ordinary
    return _hipops.BatchedLossFn.apply(lb, records, slots, [v.weight.scalar for v in order], [v.weight.per_sample for v in order], scale,
                                       len(inputs), *inputs, *[v.val for v in ordinary])


# ---------------------------------------------------------------------------------------------
# learning-rate schedules (reference :577-629)
# ---------------------------------------------------------------------------------------------
def _step_factor(i, gamma, steps):
    return gamma ** [j for j, s in enumerate([0] + list(steps)) if i > s][-1]


def LinearUpThenSteps(optimizer, num_up, gamma, steps):
    return LambdaLR(optimizer, lambda i: (i + 1) / num_up if i < num_up else _step_factor(i, gamma, steps))


def ExponentialUpThenSteps(optimizer, num_up, gamma, steps):
    """0.01 -> 1 exponentially over `num_up` epochs, then gamma^k after the k-th step epoch."""
    eps = 1.0e-2

    def factor(i):
        if i < num_up:
            return eps * math.exp(-math.log(eps) * (i + 1) / num_up)
        return _step_factor(i, gamma, steps)

    return LambdaLR(optimizer, factor)


# ---------------------------------------------------------------------------------------------
# fused clip + Adam
# ---------------------------------------------------------------------------------------------
class ClipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (betas, eps, L2 weight_decay, per-group lr, per-parameter step counts) preceded by
    clip_grad_norm_(all params, max_norm) - one C-ABI call, two kernel launches, no host sync.
    State layout matches torch.optim.Adam (`step`, `exp_avg`, `exp_avg_sq`); the step counts live on the device (the
    kernel advances them), so a step can be captured in a hipGraph and `state_dict()` / `load_state_dict()` resume
    exactly.  `grad_scale` (default 1): the stored gradients are read as grad_scale * g - 1/world when a
    data-parallel all-reduce left sums in place (trackertraincode.parallel)."""

    MAX_GROUPS = 128  # TTK_ADAM_MAX_GROUPS (prepare_finetune() of the default backbone returns 66, one per backbone sub-module)
    CHUNK = 4096  # elements per workgroup: ~800 workgroups for the 3.2 M parameters (16384 left most of the 256 CUs idle)

    def __init__(self, params, lr=1.0e-3, betas=(0.9, 0.999), eps=1.0e-8, weight_decay=0.0, max_norm: float | None = 1.0):
        # capturable: torch then keeps a loaded `step` as a float32 tensor on the parameter's device
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))
        self.param_groups = [g for g in self.param_groups]
        if len(self.param_groups) > self.MAX_GROUPS:
            raise ValueError(f"ClipAdam supports at most {self.MAX_GROUPS} parameter groups")
        if len({(g["betas"], g["eps"]) for g in self.param_groups}) != 1:
            raise ValueError("all groups must share betas and eps")
        self.max_norm = max_norm
        self.grad_scale = 1.0
        self._tables = None
        self._uploaded = None      # gradient addresses the device table currently holds
        self._upload_event = None  # recorded behind the last upload of the pinned table
        self._t = 0                # optimiser steps taken (host-side count; the per-parameter counts are state[p]["step"])
        self.last_grad_norm: Tensor | None = None

    def _invalidate(self):
        self._tables, self._uploaded, self._upload_event = None, None, None

    def load_state_dict(self, state_dict):
        """Resume: the loaded moments / step counts replace the live ones, so every cached device address is stale."""
        super().load_state_dict(state_dict)
        self._invalidate()
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        self._t = int(max(steps)) if steps else 0

    def __setstate__(self, state):
        super().__setstate__(state)
        self._invalidate()

    def _build_tables(self):
        plist, groups = [], []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("ClipAdam needs contiguous float32 CUDA parameters")
                plist.append(p)
                groups.append(gi)
        dev = plist[0].device
        # per-parameter step counts in ONE device array; state[p]["step"] are views of it (what torch.optim.Adam keeps per
        # parameter, and what state_dict() saves)
        steps = torch.zeros(len(plist), dtype=torch.float32, device=dev)
        for ti, p in enumerate(plist):
            st = self.state[p]
            if "step" in st:
                steps[ti] = float(st["step"])  # only after load_state_dict: rare
            st["step"] = steps[ti]
            for key in ("exp_avg", "exp_avg_sq"):
                if key not in st:
                    st[key] = torch.zeros_like(p)
                elif not (st[key].is_cuda and st[key].dtype == torch.float32 and st[key].is_contiguous()):
                    st[key] = st[key].to(device=dev, dtype=torch.float32).contiguous()
        ct, co = [], []
        for ti, p in enumerate(plist):
            for off in range(0, p.numel(), self.CHUNK):
                ct.append(ti)
                co.append(off)
        i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
        self._tables = dict(
            params=plist, numel=i32([p.numel() for p in plist]), group=i32(groups), chunk_tensor=i32(ct), chunk_offset=i32(co),
            nchunks=len(ct), ptrs_host=torch.zeros((len(plist), 4), dtype=torch.int64).pin_memory(),
            ptrs=torch.zeros((len(plist), 4), dtype=torch.int64, device=dev), steps=steps,
            partial=torch.empty(len(ct), dtype=torch.float32, device=dev), norm=torch.zeros(1, dtype=torch.float32, device=dev),
            hyper=torch.zeros(2 * self.MAX_GROUPS, dtype=torch.float32, device=dev),  # TTK_ADAM_HYPER_* block (include/ttk.h)
        )
        h = self._tables["ptrs_host"]
        for ti, p in enumerate(plist):
            st = self.state[p]
            h[ti, 0], h[ti, 2], h[ti, 3] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        self._uploaded = None

    def _upload_grad_pointers(self, T, capturing):
        """The device table of (param, grad, exp_avg, exp_avg_sq) addresses.  Only the gradient column ever changes, and
        only when the allocator hands out new addresses; the pinned staging table is rewritten only after the previous
        asynchronous upload has read it (a host that runs several steps ahead of the GPU would otherwise overwrite the
        addresses of a step that has not been copied yet)."""
        gptrs = []
        for p in T["params"]:
            g = p.grad
            if g is not None and not g.is_contiguous():
                g = p.grad = g.contiguous()
            gptrs.append(0 if g is None else g.data_ptr())
        gptrs = tuple(gptrs)
        if gptrs == self._uploaded:
            return
        if self._upload_event is not None and not capturing:
            self._upload_event.synchronize()
        T["ptrs_host"][:, 1] = torch.tensor(gptrs, dtype=torch.int64)
        T["ptrs"].copy_(T["ptrs_host"], non_blocking=True)
        if not capturing:
            self._upload_event = torch.cuda.Event()
            self._upload_event.record()
        self._uploaded = gptrs

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        if self._tables is None:
            self._build_tables()
        T = self._tables
        capturing = torch.cuda.is_current_stream_capturing()
        self._upload_grad_pointers(T, capturing)
        b1, b2 = self.param_groups[0]["betas"]
        G = self.MAX_GROUPS
        lr4 = (ctypes.c_float * G)(*([g["lr"] for g in self.param_groups] + [0.0] * G)[:G])
        wd4 = (ctypes.c_float * G)(*([g["weight_decay"] for g in self.param_groups] + [0.0] * G)[:G])
        p_ = _hip.ptr
        # inside a hipGraph capture nothing of the step may be baked into launch arguments: learning rates and weight
        # decays are read from the device block `hyper` (the step counts always live on the device)
        hyper = p_(T["hyper"]) if capturing else None
        _hip.lib().call("ttk_clip_adam", p_(T["ptrs"]), p_(T["numel"]), p_(T["group"]), p_(T["chunk_tensor"]), p_(T["chunk_offset"]),
                        T["nchunks"], self.CHUNK, lr4, wd4, b1, b2, self.param_groups[0]["eps"], float(self.max_norm or 0.0),
                        float(self.grad_scale), p_(T["steps"]), p_(T["partial"]), p_(T["norm"]), hyper)
        if not capturing:
            self._t += 1
        self.last_grad_norm = T["norm"]
        return None

    def _hyper_values(self):
        G = self.MAX_GROUPS
        return ([g["lr"] for g in self.param_groups] + [0.0] * G)[:G] + ([g["weight_decay"] for g in self.param_groups] + [0.0] * G)[:G]

    # ---- hipGraph support -----------------------------------------------------------------------
    def sync_hyper_to_device(self):
        """Write the groups' lr / weight_decay into the device block a captured step reads.  Called before a capture and
        whenever the scheduler changed a learning rate (once per epoch)."""
        if self._tables is None:
            self._build_tables()
        vals = self._hyper_values()
        self._tables["hyper"].copy_(torch.tensor(vals, dtype=torch.float32))  # synchronous, pageable: rare
        self._hyper_sig = tuple(vals)

    def before_graph_replay(self):
        """Push a changed learning rate / weight decay (scheduler step) to the device block before the replay."""
        vals = tuple(self._hyper_values())
        if vals != getattr(self, "_hyper_sig", None):
            torch.cuda.current_stream().synchronize()  # earlier replays still read the old values
            self.sync_hyper_to_device()

    def after_graph_replay(self):
        """Host-side bookkeeping of one replayed step (the device counters were advanced by the graph)."""
        self._t += 1


# ---------------------------------------------------------------------------------------------
# stochastic weight averaging (reference SwaCallback :447-467)
# ---------------------------------------------------------------------------------------------
class SwaCallback:
    """Equal-weight running average of parameters AND buffers (AveragedModel(use_buffers=True)),
    updated once per epoch for epochs > start_epoch, kept on the CPU, saved as swa.ckpt."""

    def __init__(self, start_epoch):
        self._start_epoch = start_epoch
        self._swa_model = None
        self.n_averaged = 0

    @property
    def swa_model(self):
        return self._swa_model

    def on_train_start(self, model: nn.Module):
        import copy

        self._swa_model = copy.deepcopy(model).to("cpu")

    @torch.no_grad()
    def on_train_epoch_end(self, epoch: int, model: nn.Module):
        if epoch <= self._start_epoch:
            return
        avg, new = self._swa_model.state_dict(), model.state_dict()
        for k, a in avg.items():
            b = new[k].detach().to("cpu")
            if self.n_averaged == 0:
                a.copy_(b)
            elif a.is_floating_point():
                a.add_((b - a) / (self.n_averaged + 1))
            else:
                a.copy_(a + torch.div(b - a, self.n_averaged + 1, rounding_mode="trunc"))
        self.n_averaged += 1

    def on_train_end(self, root_dir: str):
        assert self._swa_model is not None
        save_model(self._swa_model, os.path.join(root_dir, "swa.ckpt"))


# ---------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------
def _concat_rows(tensors):
    """torch.concat(tensors, dim=0) - as a VIEW when the tensors already lie back to back in one allocation (the static inputs of
    GraphedTrainStep are laid out that way: at B = 512 the copy of the images alone is 34 MB read + 34 MB written per step)."""
    t0 = tensors[0]
    if len(tensors) == 1:
        return t0
    adjacent = all(t.is_contiguous() and not t.requires_grad and t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:] for t in tensors)
    if adjacent:
        base = t0.untyped_storage().data_ptr()
        end = t0.data_ptr()
        for t in tensors:
            adjacent = adjacent and t.untyped_storage().data_ptr() == base and t.data_ptr() == end
            end = t.data_ptr() + t.numel() * t.element_size()
    if not adjacent:
        return torch.concat(tensors, dim=0)
    rows = sum(int(t.shape[0]) for t in tensors)
    return torch.as_strided(t0, (rows,) + tuple(t0.shape[1:]), t0.stride())


def training_step(model: nn.Module, batches: List[Batch], epoch: int, criterions):
    """LitModel.training_step (scripts/train_poseestimator.py:310-330) without the logging."""
    if batches[0]["image"].is_cuda:
        _hip.lib().clear_stale_error("the start of a training step")  # once per step (the launches themselves no longer do it)
    inputs = _concat_rows([b["image"] for b in batches])
    ids = _concat_rows([b["coord_convention_id"] for b in batches])
    preds = model(inputs, ids)
    loss_sum, all_lossvals = default_compute_loss(preds, batches, epoch, criterions)
    by_name = concatenated_values_by_name(itertools.chain.from_iterable(all_lossvals))
    return {"loss": loss_sum, "mt_losses": by_name}


def _criterion_weights(c, step):
    """Flat tuple of every weight in a criterion tree at `step` (they are constants inside a captured graph)."""
    if isinstance(c, dict):
        return tuple((str(k), _criterion_weights(v, step)) for k, v in c.items())
    if isinstance(c, CriterionGroup):
        return (_weight_at(c.w, step),) + tuple(_criterion_weights(x, step) for x in c.criterions)
    return (_weight_at(c.w, step),)


class GraphedTrainStep:
    """zero_grad + training_step + backward + ClipAdam.step captured ONCE as a hipGraph and replayed every step.

    A pose-estimator step is ~200 kernel launches, half of them small head / loss / bookkeeping kernels; enqueueing
    them from Python costs ~4.7 ms of host time per step next to ~8.8 ms of GPU work (B = 512) - on a slower host, or
    at smaller batches, that host time is the bound and a captured graph, which enqueues the same work with one call,
    removes it.  (The reference has no counterpart: Lightning drives eager PyTorch; `train_poseestimator.py:442-454`.)

    The graph stays valid while the sub-batch layout (tags, sizes, fields), the epoch-dependent loss weights and the
    learning-rate-independent launch arguments stay the same: `run()` compares a signature and re-captures when it
    changes (e.g. during the NLL ramp epochs).  Learning rates, weight decays and Adam's step count live in device
    memory (ClipAdam.sync_hyper_to_device), so scheduler steps need no re-capture.  New batches are copied into the
    graph's static input tensors.  Single-GPU: the data-parallel all-reduce is issued eagerly (train.fit)."""

    def __init__(self, model: nn.Module, criterions, optimizer: "ClipAdam"):
        if not isinstance(optimizer, ClipAdam):
            raise TypeError("GraphedTrainStep needs the fused ClipAdam optimiser (its step is capturable)")
        self.model, self.criterions, self.optimizer = model, criterions, optimizer
        self.graph = None
        self._sig = None
        self._static: list[Batch] = []
        self._out = None
        self.captures = 0
        self._miss_streak = 0      # consecutive steps whose signature differed from the captured one
        self.eager_only = False    # set once re-capturing is seen to happen step after step

    def _signature(self, batches, epoch):
        layout = tuple((str(b.meta.tag), b.meta.batchsize, tuple((k, tuple(v.shape), str(v.dtype)) for k, v in b.items()))
                       for b in batches)
        return layout, _criterion_weights(self.criterions, epoch), self.model.training

    def _eager(self, batches, epoch):
        self.optimizer.zero_grad(set_to_none=True)
        out = training_step(self.model, batches, epoch, self.criterions)
        out["loss"].backward()
        self.optimizer.step()
        return out

    def _capture(self, batches, epoch):
        # static inputs: the fields every sub-batch has (image, coord_convention_id, ...) are views of ONE tensor per field, in sub-batch
        # order, so that training_step's concatenation over the sub-batches is a view (_concat_rows) instead of a copy per step
        shared = set.intersection(*[set(k for k, v in b.items() if torch.is_tensor(v)) for b in batches]) if len(batches) > 1 else set()
        joined = {k: torch.concat([b[k] for b in batches], dim=0).split([int(b[k].shape[0]) for b in batches], dim=0) for k in shared
                  if all(b[k].dim() >= 1 and b[k].shape[1:] == batches[0][k].shape[1:] and b[k].dtype == batches[0][k].dtype for b in batches)}
        self._static = [Batch(b.meta, ((k, joined[k][i] if k in joined else v.clone()) for k, v in b.items())) for i, b in enumerate(batches)]
        self.optimizer.sync_hyper_to_device()
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)  # gradients are allocated from the graph's pool at fixed addresses
        with _hip.CAPTURE_LOCK:  # (a loader's prefetch thread must not allocate or copy while the capture is open: _hip.CAPTURE_LOCK)
            with torch.cuda.graph(self.graph):
                out = training_step(self.model, self._static, epoch, self.criterions)
                out["loss"].backward()
                self.optimizer.step()
        self._out = out
        self.captures += 1

    def run(self, batches: List[Batch], epoch: int):
        """One training step.  Returns {"loss", "mt_losses"}; when replayed these are the graph's static output
        tensors (overwritten by the next call)."""
        if self.eager_only:
            return self._eager(batches, epoch)
        sig = self._signature(batches, epoch)
        if sig != self._sig:
            # A loader that mixes datasets of several Tags draws the per-Tag sub-batch sizes anew every step (ResidentLoader): the signature
            # then changes almost every step and each step would run eagerly AND re-capture (synchronise, capture, a new private pool) -
            # far slower than eager.  Three misses in a row: stay eager for the rest of the run (round-3 advisor finding).
            self._miss_streak += 1
            if self._miss_streak >= 3 and self.captures >= 2:
                import warnings
                warnings.warn("GraphedTrainStep: the sub-batch layout changed on three consecutive steps (per-Tag batch sizes vary from step to "
                              "step?) - a captured graph would be re-captured every step; running eagerly from here on", RuntimeWarning, stacklevel=2)
                self.eager_only, self.graph, self._static, self._out = True, None, [], None
                return self._eager(batches, epoch)
            # first step with this layout: run it eagerly (lazy initialisation, table building, stream creation happen
            # here and the step counts), then capture for the following steps
            out = self._eager(batches, epoch)
            # hand back detached copies and drop the eager autograd graph BEFORE capturing: with it still alive,
            # hipStreamEndCapture / graph instantiation was seen to segfault on ROCm 7.2 (tools/debug/graph_capture4.py)
            out = {"loss": out["loss"].detach().clone(), "mt_losses": {k: v.detach().clone() for k, v in out["mt_losses"].items()}}
            torch.cuda.synchronize()
            self._capture(batches, epoch)
            self._sig = sig
            return out
        # new batches -> the graph's static inputs: all float32 tensors in one launch (ttk_multi_copy), the rest (ids) one by one
        fsrc, fdst = [], []
        for dst, src in zip(self._static, batches):
            for k, v in src.items():
                d = dst[k]
                if v is d or (torch.is_tensor(v) and v.data_ptr() == d.data_ptr() and v.numel() == d.numel()):
                    continue  # the loader already wrote into the static tensor
                if (torch.is_tensor(v) and v.is_cuda and v.dtype == d.dtype and v.element_size() % 4 == 0 and v.is_contiguous() and d.is_contiguous()
                        and v.numel() == d.numel() and v.dim() >= 1 and v.data_ptr() % 4 == 0):
                    # moved as 32-bit words whatever the dtype (int64 ids, float32 labels): one launch for all of them
                    fsrc.append(v if v.dtype == torch.float32 else v.view(torch.float32))
                    fdst.append(d if d.dtype == torch.float32 else d.view(torch.float32))
                else:
                    d.copy_(v, non_blocking=True)
        if fdst:
            _hip.lib().multi_copy(fsrc, fdst)
        self._miss_streak = 0
        self.optimizer.before_graph_replay()
        self.graph.replay()
        self.optimizer.after_graph_replay()
        return self._out


@torch.no_grad()
def validate(model: nn.Module, val_loader, val_criterions) -> float:
    """One validation epoch with LitModel.validation_step's arithmetic (scripts/train_poseestimator.py:332-338) and Lightning's
    epoch reduction of `self.log("val_loss", ..., on_epoch=True, batch_size=n)`: per batch the SUM over samples and terms of
    value * weight (not a mean), per epoch the batch-size-weighted mean of those sums.  The model runs in eval mode without
    coord_convention_id; the criterions receive the BATCH INDEX as their step (the reference's quirk: weights that ramp with the
    epoch ramp with the batch index here)."""
    was_training = model.training
    model.eval()
    _hip.lib().clear_stale_error("validate")  # (once per validation epoch: a pending error of an unrelated earlier call must not be blamed on these launches)
    total, count = None, 0
    try:
        for batch_idx, batch in enumerate(val_loader):
            if isinstance(batch, (list, tuple)):
                raise TypeError("validate: the validation loader yields ONE Batch per iteration (reference pipelines.py:543-552), got a list - "
                                "train loaders yield lists")
            pred = model(batch["image"])
            crit = val_criterions[batch.meta.tag] if isinstance(val_criterions, dict) else val_criterions
            values = crit.evaluate(pred, batch, batch_idx)
            val_loss = torch.cat([(lv.val * lv.weight).reshape(-1) for lv in values]).sum()
            n = int(batch.meta.batchsize)
            total = val_loss * n if total is None else total + val_loss * n
            count += n
    finally:
        model.train(was_training)
    if count == 0:
        raise ValueError("empty validation loader")
    return float(total.item()) / count


class CheckpointCallback:
    """ModelCheckpoint(save_top_k=1, save_last=True, monitor="val_loss", filename="best") of the reference's training script
    (:423-431) followed by its final re-save in the plain format (:460-465): after every validation epoch `last.ckpt` is written,
    and `best.ckpt` whenever the monitored value reaches a new minimum - both with `save_model` (state dict + constructor
    arguments, loadable by `models.load_model`)."""

    def __init__(self, dirpath: str):
        self.dirpath = dirpath
        self.best_value = math.inf
        self.best_epoch = -1
        self.history: list[float] = []

    @property
    def best_model_path(self):
        return os.path.join(self.dirpath, "best.ckpt")

    @property
    def last_model_path(self):
        return os.path.join(self.dirpath, "last.ckpt")

    def _save(self, model, path):
        import copy

        os.makedirs(self.dirpath, exist_ok=True)
        save_model(copy.deepcopy(model).to("cpu"), path)

    def on_validation_end(self, epoch: int, model: nn.Module, val_loss: float):
        self.history.append(val_loss)
        self._save(model, self.last_model_path)
        if val_loss < self.best_value:
            self.best_value, self.best_epoch = val_loss, epoch
            self._save(model, self.best_model_path)


def fit(model: nn.Module, train_loader, criterions, optimizer, scheduler=None, epochs=1, callbacks=(), on_step=None,
        grad_sync=None, val_loader=None, val_criterions=None, reducer=None, graphed=False):
    """Epoch loop with Lightning's ordering: per step zero_grad -> forward -> loss -> backward -> (gradient exchange) -> clip + Adam;
    per epoch the scheduler step, then - when `val_loader` is given - a validation epoch (`validate`) whose value goes to the
    callbacks' `on_validation_end(epoch, model, val_loss)` (CheckpointCallback: best.ckpt / last.ckpt), then `on_train_epoch_end`.
    Data-parallel replicas pass their `parallel.GradAllReduce` as `reducer`: `begin_step()` before the step, `finish()` between
    backward and the optimiser (it waits for the in-place all-reduces that ran during backward), and - should backward raise -
    `abort()` so that no collective is left in flight on gradient memory that is about to be freed.  `grad_sync(model)` is the
    older hook form of the same (runs between backward and the optimiser step).
    Gradients are dropped (set_to_none) before every step: the arena views autograd installs are the exchange buffers.
    `graphed=True` (single replica, ClipAdam): every step is a replay of ONE captured hipGraph (`GraphedTrainStep`; re-captured when the
    sub-batch layout or the epoch's loss weights change); `on_step` then receives the graph's static output tensors."""
    stepper = None
    if graphed:
        if reducer is not None or grad_sync is not None:
            raise ValueError("graphed steps are single-replica (collectives inside a captured graph are untested on this stack)")
        stepper = GraphedTrainStep(model, criterions, optimizer)
    for cb in callbacks:
        if hasattr(cb, "on_train_start"):
            cb.on_train_start(model)
    model.train()
    params = list(model.parameters()) if reducer is not None else None
    for epoch in range(epochs):
        for batches in train_loader:
            if stepper is not None:
                out = stepper.run(batches, epoch)
                if on_step is not None:
                    on_step(epoch, out)
                continue
            optimizer.zero_grad(set_to_none=True)
            if reducer is not None:
                reducer.begin_step()
            out = training_step(model, batches, epoch, criterions)
            try:
                out["loss"].backward()
            except BaseException:
                if reducer is not None:
                    reducer.abort()
                raise
            if reducer is not None:
                reducer.finish(params)
            if grad_sync is not None:
                grad_sync(model)
            optimizer.step()
            if on_step is not None:
                on_step(epoch, out)
        if scheduler is not None:
            scheduler.step()
        if val_loader is not None:
            val_loss = validate(model, val_loader, val_criterions if val_criterions is not None else criterions)
            for cb in callbacks:
                if hasattr(cb, "on_validation_end"):
                    cb.on_validation_end(epoch, model, val_loss)
        for cb in callbacks:
            if hasattr(cb, "on_train_epoch_end"):
                cb.on_train_epoch_end(epoch, model)
    return model
