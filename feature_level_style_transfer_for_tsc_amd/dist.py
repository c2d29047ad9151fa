"""Data parallelism for the train step: one process per GPU, one flat fp32 gradient bucket per step,
all-reduced over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).

The whole model is ≈ 8.9 M parameters ≈ 35.6 MB, so a single bucket (one collective per step) is the
right shape for point-to-point xGMI links: per-link time ≈ 2·(N−1)/N·35.6 MB / 153 GB/s ≈ 0.4 ms.
The step's only other collectives are two tiny scalar averages that keep GradNorm identical on every
rank.  Two modes:

* **A, "DDP semantics"** (default; the benchmark): batch-coupled statistics (BatchNorm batch moments, CPC
  negatives, NoiseTransfer means, CDAN's batch sums) stay per-rank; collectives = the bucket + 10 scalars,
  outside the captured hipGraphs.
* **B, "global-batch exact"** (``with global_batch(bucket):`` / ``JointTrainer(..., sync="global")``): every
  batch-coupled quantity is formed over the samples of ALL ranks, so N ranks × B/N samples reproduce the
  single-process step on the B-sample batch: SyncBN (moments and the two backward means all-reduced), CPC scored
  against the predictions gathered from every rank, NoiseTransfer means and CDAN's batch sums all-reduced,
  GradNorm's per-loss gradients averaged before their norms.  Eager only (collectives sit inside autograd).

Gradient convention in both modes: every rank differentiates the mean over ITS samples; the bucket averages the
parameter gradients.  A collective that mixes ranks inside the graph is therefore an autograd function whose
backward is the matching collective (all-reduce-mean ↔ all-reduce-mean, gather ↔ sum-scatter).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradBucket:
    def __init__(self, process_group: Optional[dist.ProcessGroup] = None, always_reduce: bool = False):
        """``always_reduce``: issue the collectives even in a one-rank group (a no-op numerically) — lets a one-GPU box
        execute the RCCL code path end to end (tests/test_gpu_dist.py)."""
        if not dist.is_initialized():
            raise RuntimeError("GradBucket needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.always_reduce = always_reduce
        self._flat: Optional[torch.Tensor] = None
        self._stream = None           # side stream of the bucket all-reduce (GPU only)
        self._pending = False

    def _buffer(self, n: int, like: torch.Tensor) -> torch.Tensor:
        if self._flat is None or self._flat.numel() != n or self._flat.device != like.device:
            self._flat = torch.empty(n, device=like.device, dtype=torch.float32)
        return self._flat

    def all_reduce(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Average ``.grad`` of every parameter that has one, in place, with ONE collective."""
        self.all_reduce_begin(params)
        self.all_reduce_end()

    def all_reduce_begin(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Start averaging ``.grad`` of every parameter that has one.  On the GPU the pack, the collective, the scale and the
        copy back run on a SIDE stream ordered after everything already issued on the current stream, so work issued next on
        the current stream that does not touch ``.grad`` (GradNorm's partial backward passes) overlaps the all-reduce;
        ``all_reduce_end`` orders the current stream after it."""
        self._pending = False
        grads: List[torch.Tensor] = [p.grad for p in params if p.grad is not None]
        if not grads or (self.world == 1 and not self.always_reduce):
            return
        n = sum(g.numel() for g in grads)
        flat = self._buffer(n, grads[0])
        views, off = [], 0
        for g in grads:
            views.append(flat[off: off + g.numel()].view_as(g))
            off += g.numel()

        def body():
            torch._foreach_copy_(views, grads)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.mul_(1.0 / self.world)
            torch._foreach_copy_(grads, views)
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                body()
            self._pending = True
        else:
            body()

    def all_reduce_end(self) -> None:
        if getattr(self, "_pending", False):
            torch.cuda.current_stream().wait_stream(self._stream)
            self._pending = False

    def mean_scalars(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.always_reduce:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world


def shard_batch(n_items: int, rank: int, world: int) -> slice:
    """Contiguous, equal shard of a global batch (remainder items go to the lowest ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))


# --------------------------------------------------------------------------------------------------
# mode B: global-batch exact
# --------------------------------------------------------------------------------------------------
_GLOBAL: Optional[GradBucket] = None


class global_batch:
    """Context: batch-coupled ops executed inside (forward AND backward) reduce over every rank of ``bucket``."""

    def __init__(self, bucket: Optional[GradBucket]):
        self.bucket = bucket

    def __enter__(self):
        global _GLOBAL
        self._prev, _GLOBAL = _GLOBAL, (self.bucket if self.bucket is not None and self.bucket.world > 1 else None)
        return self

    def __exit__(self, *exc):
        global _GLOBAL
        _GLOBAL = self._prev
        return False


def global_batch_active() -> bool:
    return _GLOBAL is not None


def world() -> int:
    return _GLOBAL.world if _GLOBAL is not None else 1


def rank() -> int:
    return dist.get_rank(_GLOBAL.group) if _GLOBAL is not None else 0


def sum_over_ranks_(t: torch.Tensor) -> int:
    """In-place sum of a (gradient-free) tensor over the ranks of the active global batch; returns the world size
    (1 and no-op outside the context)."""
    if _GLOBAL is None:
        return 1
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_GLOBAL.group)
    return _GLOBAL.world


def _all_gather_cat(x: torch.Tensor, dim: int, group) -> torch.Tensor:
    """Every rank's ``x`` concatenated along ``dim`` in rank order (no gradient).  RCCL: one all_gather_into_tensor of the
    payload; other backends (gloo in the CPU tests): own block + all-reduce(SUM) of a zero-padded buffer."""
    n, r = dist.get_world_size(group), dist.get_rank(group)
    x = x.detach().contiguous()
    if dist.get_backend(group) == "nccl":
        out = x.new_empty((n,) + tuple(x.shape))
        dist.all_gather_into_tensor(out, x, group=group)
        return torch.cat(list(out.unbind(0)), dim=dim) if dim != 0 else out.view((n * x.shape[0],) + tuple(x.shape[1:]))
    shape = list(x.shape)
    b = shape[dim]
    shape[dim] = n * b
    out = x.new_zeros(shape)
    out.narrow(dim, r * b, b).copy_(x)
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


def gather_slots(part: torch.Tensor) -> torch.Tensor:
    """BatchNorm moment partials [C, slots, 4] of every rank as [C, world·slots, 4] (identity outside the context)."""
    return part if _GLOBAL is None else _all_gather_cat(part, 1, _GLOBAL.group)


class _AllReduceMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = x.detach().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=_GLOBAL.group)
        ctx.group, ctx.n = _GLOBAL.group, _GLOBAL.world
        return y / _GLOBAL.world

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g / ctx.n


def mean_over_ranks(x: torch.Tensor) -> torch.Tensor:
    """Differentiable mean over ranks (identity outside the context).  Every rank's loss depends on the result, and
    every rank's gradient is averaged afterwards, so the backward is again the mean over ranks."""
    return x if _GLOBAL is None else _AllReduceMean.apply(x)


class _GatherCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dim):
        out = _all_gather_cat(x, dim, _GLOBAL.group)    # RCCL: all_gather_into_tensor (1/N of the padded all-reduce's bytes)
        ctx.group, ctx.dim, ctx.b, ctx.r = _GLOBAL.group, dim, x.shape[dim], dist.get_rank(_GLOBAL.group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()                      # every rank holds d(its loss)/d(all blocks): sum, keep own block
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g.narrow(ctx.dim, ctx.r * ctx.b, ctx.b).contiguous(), None


def gather_cat(x: torch.Tensor, dim: int) -> torch.Tensor:
    """Differentiable concatenation of every rank's ``x`` along ``dim`` in rank order (identity outside the context);
    backward = sum over ranks of the incoming gradients, own block kept."""
    return x if _GLOBAL is None else _GatherCat.apply(x, dim)
