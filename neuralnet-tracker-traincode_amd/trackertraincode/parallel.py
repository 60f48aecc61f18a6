"""Data-parallel replicas over RCCL/xGMI (no counterpart in the reference, which is single-device:
SURVEY.md §2.2).  One process per GPU; every rank holds a full replica (12.9 MB of fp32 parameters),
crops are sharded over ranks, BatchNorm statistics stay per-replica (what stock DDP would do to the
reference), and the only exchange step is the gradient all-reduce.

Zero-copy exchange.  The backbone's backward writes every parameter gradient into ONE flat arena and
announces finished ranges of it in reverse layer order (`grad_ready_hook(arena, entries)`, largest
tensors first: dw6 + dw5_6 = 6.3 MB are final after ~15 % of backward); the fused heads do the same with
their own small arena.  `GradAllReduce` merges adjacent ranges into buckets of >= `bucket_bytes` and
all-reduces each bucket IN PLACE, asynchronously, on a side stream (torch.distributed "nccl" backend =
RCCL on ROCm) while backward continues - no packing copy before the collective, none after it: the
tensors autograd installs as `param.grad` are views of the arena.  The sums are not divided by the
world size here; `grad_scale` (= 1/world) is handed to the fused clip+Adam kernel, which reads every
gradient as grad_scale * g (train.ClipAdam.grad_scale).  `finish()` issues O(buckets) launches.

Gradients that did not come through an arena (a handful of tiny tensors such as DiagonalScaleParameter.hidden_scale)
are packed into one extra bucket by `finish()`.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of 12.9 MB moves ~22.6 MB per
GPU, ~0.15 ms - far below a step, so a handful of large buckets is the right granularity.
"""
from __future__ import annotations

from typing import Iterable, Sequence

import time

import torch
import torch.distributed as dist


class GradAllReduce:
    def __init__(self, process_group=None, bucket_bytes: int = 4 << 20, always_reduce: bool = False):
        """`always_reduce`: issue the collectives even in a world of one (they are identities) - used to exercise the
        stream / event / bucket logic on a single GPU."""
        self.pg = process_group
        # Diagnostics (bench.py's `comm` object): with `measure` set, every collective is bracketed by events on the side stream (the
        # stream then waits for each collective itself, so the buckets of a step serialise there - as they do inside RCCL) and finish()
        # brackets the wait of the main stream; collect_timing() returns the sums.
        # Off in the timed region of a benchmark: the extra waits and events are not free.
        self.measure = False
        self._timing: list = []
        self._finish_log: list = []
        self.last_bucket_bytes: list = []             # sizes of the collectives of the last finished step (bench.py --comm-only)
        self._bucket_bytes_now: list = []
        self._timing_mark = 0  # len(_timing) at the end of the last completed step
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._always = always_reduce
        self.bucket_bytes = bucket_bytes
        self._stream = None
        self._works: list = []                      # collectives in flight
        self._open = None                           # (arena, lo, hi) still growing towards a bucket
        self._done: list[tuple[torch.Tensor, int, int]] = []  # ranges already handed to RCCL this step
        self._entries: list[tuple[torch.nn.Parameter, torch.Tensor, int, int]] = []  # (param, arena, lo, hi)
        self.collectives = 0                        # issued in the current / last step (tests, launch-count evidence)
        self.zero_copy = self.copied = 0            # parameters whose .grad was / was not the arena view itself (last finish())

    @property
    def grad_scale(self) -> float:
        """What the optimiser must multiply the exchanged gradients with (they are sums over the replicas)."""
        return 1.0 / self.world

    @property
    def active(self) -> bool:
        return self.world > 1 or self._always

    # ---- called during backward (autograd worker thread) ---------------------------------------------------------
    def on_ready(self, arena: torch.Tensor, entries: Sequence[tuple]):
        """`entries` = [(param, lo, hi), ...]: elements [lo, hi) of the flat float32 `arena` hold the final gradient of
        `param`; the entries of one call are adjacent (padding between tensors belongs to the range)."""
        if not self.active or not entries:
            return
        lo, hi = min(e[1] for e in entries), max(e[2] for e in entries)
        self._entries += [(p, arena, a, b) for p, a, b in entries]
        if self._open is not None:
            o_arena, o_lo, o_hi = self._open
            if o_arena is arena and lo <= o_hi and hi >= o_lo:
                self._open = (arena, min(lo, o_lo), max(hi, o_hi))  # adjacent (reverse layer order: the new range ends where the open one begins)
            else:
                self._flush()
                self._open = (arena, lo, hi)
        else:
            self._open = (arena, lo, hi)
        _, o_lo, o_hi = self._open
        if (o_hi - o_lo) * 4 >= self.bucket_bytes:
            self._flush()

    def _all_reduce(self, flat: torch.Tensor):
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))  # the producing kernels of this range are enqueued
            with torch.cuda.stream(self._stream):
                if self.measure:  # (diagnostic pass only: the host-side wait below serialises backward under gloo - bench.py's `comm.note` says so)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    t0 = time.perf_counter()
                    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                    work.wait()  # RCCL: the side stream waits for the collective (the host does not); gloo: the host does
                    host_ms = (time.perf_counter() - t0) * 1e3
                    e1.record()
                    self._timing.append((flat.numel() * flat.element_size(), e0, e1, host_ms))
                else:
                    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._works.append(work)
        self.collectives += 1
        self._bucket_bytes_now.append(flat.numel() * flat.element_size())

    def _flush(self):
        if self._open is None:
            return
        arena, lo, hi = self._open
        self._open = None
        self._all_reduce(arena[lo:hi])  # a view: the collective works in place
        self._done.append((arena, lo, hi))

    # ---- called after backward, before the optimiser -------------------------------------------------------------
    def finish(self, params: Iterable[torch.nn.Parameter] = ()):
        """Reduce what is still open, plus the gradients of `params` that no arena covered, and make the current stream
        wait for every collective.  Afterwards each `param.grad` holds the SUM over the replicas."""
        if not self.active:
            return
        self._flush()
        covered = {id(p) for p, _, _, _ in self._entries}
        rest = [p for p in params if p.grad is not None and id(p) not in covered]
        packed = None
        if rest:
            packed = torch.cat([p.grad.reshape(-1) for p in rest])  # a handful of tiny tensors: one launch
            self._all_reduce(packed)
        ev_bwd = ev_done = None
        if self.measure and self._stream is not None:
            ev_bwd, ev_done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev_bwd.record()  # fires when the last kernel of backward is done
        for w in self._works:
            w.wait()  # CUDA: the current stream waits for the collective
        if self.measure and self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
            ev_done.record()  # fires when the optimiser may start: backward done AND every collective done
            self._finish_log.append((ev_bwd, ev_done))
        if packed is not None:
            off = 0
            for p in rest:
                n = p.grad.numel()
                p.grad.copy_(packed[off:off + n].view_as(p.grad))
                off += n
        # autograd normally installs the arena views themselves as .grad (zero copy); if it cloned one (the view was
        # still referenced elsewhere when AccumulateGrad ran) the clone may hold pre-exchange values: overwrite it
        self.zero_copy = self.copied = 0
        for p, arena, lo, hi in self._entries:
            g = p.grad
            if g is None:
                continue
            base = arena.data_ptr() + 4 * lo
            if g.data_ptr() != base:
                g.copy_(arena[lo:lo + g.numel()].view_as(g))
                self.copied += 1
            else:
                self.zero_copy += 1
        self._works.clear()
        self._done.clear()
        self._entries.clear()
        self.last_bucket_bytes, self._bucket_bytes_now = self._bucket_bytes_now, []
        self._timing_mark = len(self._timing)  # timing records up to here belong to completed steps (abort() drops the rest)

    def begin_step(self):
        self.collectives = 0

    def collect_timing(self) -> dict:
        """After a synchronize: the diagnostics of the steps run with `measure` set since the last call - per step averages over the
        recorded finish() calls are left to the caller, this returns the raw sums: {buckets, bytes, allreduce_ms_sum, exposed_ms,
        steps}.  `allreduce_ms_sum`: device time between the moment a bucket's range was final and its collective's completion, summed over
        the buckets (they overlap backward: this is not time lost); `exposed_ms`: how long the optimiser had to wait for the exchange
        after the LAST kernel of backward had finished (time lost to communication)."""
        torch.cuda.synchronize()
        out = dict(buckets=len(self._timing), bytes=sum(t[0] for t in self._timing), allreduce_ms_sum=0.0, host_wait_ms_sum=sum(t[3] for t in self._timing),
                   exposed_ms=0.0)
        for _, e0, e1, _ in self._timing:
            out["allreduce_ms_sum"] += e0.elapsed_time(e1)
        for a, b in self._finish_log:
            out["exposed_ms"] += a.elapsed_time(b)
        out["steps"] = len(self._finish_log)
        self._timing.clear()
        self._finish_log = []
        self._timing_mark = 0
        return out

    def abort(self):
        """Backward raised: wait for the collectives already issued (they work in place on gradient arenas that are about to be
        freed) and forget the step."""
        for w in self._works:
            try:
                w.wait()
            except Exception:  # noqa: BLE001 - the original exception is the one to report
                pass
        self._works.clear()
        self._done.clear()
        self._entries.clear()
        self._open = None
        # the diagnostics of the aborted step too: a later finish() must not publish its buckets as `last_bucket_bytes`, and collect_timing()
        # must not count collectives whose step never finished (events recorded for them are dropped with the lists)
        if self._bucket_bytes_now:
            self._bucket_bytes_now = []
        # keep the records of COMPLETED steps (collect_timing() divides by len(_finish_log)); the aborted step's buckets were appended after the
        # last completed step's finish(), i.e. beyond _timing_mark
        del self._timing[self._timing_mark:]


def install(reducer: GradAllReduce | None):
    """Point the backbones' and the fused heads' gradient-ready hooks at `reducer` (None: remove them)."""
    from .backbones import mobilenet_v1, resnet
    from .neuralnets import _hipops

    hook = reducer.on_ready if reducer is not None else None
    mobilenet_v1.grad_ready_hook = hook
    resnet.grad_ready_hook = hook
    _hipops.grad_ready_hook = hook


def broadcast_module_state(module: torch.nn.Module, src: int = 0, process_group=None):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous share [begin, end) of `total` units for `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
