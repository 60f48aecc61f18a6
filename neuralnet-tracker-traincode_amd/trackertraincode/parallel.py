"""Data-parallel replicas over RCCL/xGMI (no counterpart in the reference, which is single-device:
SURVEY.md §2.2).  One process per GPU; every rank holds a full replica (12.9 MB of fp32 parameters),
crops are sharded over ranks, BatchNorm statistics stay per-replica (what stock DDP would do to the
reference), and the only exchange step is the gradient all-reduce:

  * the backbone's backward hands finished gradient groups to `GradAllReduce.on_ready` in reverse
    layer order (largest tensors first: dw6 + dw5_6 = 6.3 MB become ready after ~15 % of backward),
  * each group is packed into one flat buffer and all-reduced (sum) asynchronously on a side stream
    (torch.distributed "nccl" backend = RCCL on ROCm), overlapping the rest of backward,
  * `finish()` waits, scales by 1/world and unpacks into the gradient tensors autograd returns.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of 12.9 MB moves ~22.6 MB per
GPU, ~0.15 ms - far below a step, so a handful of large buckets is the right granularity.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class GradAllReduce:
    def __init__(self, process_group=None, bucket_bytes: int = 4 << 20, always_reduce: bool = False):
        """`always_reduce`: issue the collectives even in a world of one (they are identities) - used to exercise the
        stream / event / bucket logic on a single GPU."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._always = always_reduce
        self.bucket_bytes = bucket_bytes
        self._pending: list[tuple[torch.Tensor, list[torch.Tensor], object]] = []
        self._queue: list[torch.Tensor] = []
        self._queued_bytes = 0
        self._stream = None

    # ---- called during backward (autograd worker thread) with (param, grad) pairs whose values are final.
    # The parameter is kept because autograd may CLONE the returned gradient into param.grad (it does
    # when somebody else - like this object - still references the tensor): finish() writes to param.grad.
    def on_ready(self, pairs):
        if self.world == 1 and not self._always:
            return
        for p, g in pairs:
            if g is None:
                continue
            self._queue.append((p, g))
            self._queued_bytes += g.numel() * g.element_size()
        if self._queued_bytes >= self.bucket_bytes:
            self._flush()

    def _flush(self):
        if not self._queue:
            return
        grads, self._queue, self._queued_bytes = self._queue, [], 0
        flat = torch.cat([g.reshape(-1) for _, g in grads])
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._stream):
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            flat.record_stream(self._stream)
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append((flat, grads, work))

    # ---- called after backward, before the optimiser
    def finish(self, extra_params: Iterable[torch.nn.Parameter] = ()):
        """Reduce whatever is still queued (plus the .grad of `extra_params`, e.g. the head parameters),
        wait for all buckets and write the averaged values into the parameters' gradients."""
        if self.world == 1 and not self._always:
            return
        seen = {id(p) for _, grads, _ in self._pending for p, _ in grads} | {id(p) for p, _ in self._queue}
        self.on_ready([(p, p.grad) for p in extra_params if p.grad is not None and id(p) not in seen])
        self._flush()
        inv = 1.0 / self.world
        for flat, grads, work in self._pending:
            work.wait()  # CUDA: makes the current stream wait for the collective
            flat.mul_(inv)
            off = 0
            for p, g in grads:
                n = g.numel()
                target = p.grad if (p is not None and p.grad is not None) else g
                target.copy_(flat[off:off + n].view_as(target))
                off += n
        self._pending.clear()


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
