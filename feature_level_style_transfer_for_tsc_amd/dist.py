"""Data parallelism for the train step: one process per GPU, one flat fp32 gradient bucket per step,
all-reduced over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in the CPU tests).

The whole model is ≈ 8.9 M parameters ≈ 35.6 MB, so a single bucket (one collective per step) is the
right shape for point-to-point xGMI links: per-link time ≈ 2·(N−1)/N·35.6 MB / 153 GB/s ≈ 0.4 ms.
The step's only other collectives are two tiny scalar averages that keep GradNorm identical on every
rank.  Batch-coupled statistics (BatchNorm batch moments, CPC negatives, NoiseTransfer means) stay
per-rank — "DDP semantics" (SURVEY §8e mode A).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradBucket:
    def __init__(self, process_group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            raise RuntimeError("GradBucket needs an initialised torch.distributed process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._flat: Optional[torch.Tensor] = None

    def _buffer(self, n: int, like: torch.Tensor) -> torch.Tensor:
        if self._flat is None or self._flat.numel() != n or self._flat.device != like.device:
            self._flat = torch.empty(n, device=like.device, dtype=torch.float32)
        return self._flat

    def all_reduce(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Average ``.grad`` of every parameter that has one, in place, with ONE collective."""
        grads: List[torch.Tensor] = [p.grad for p in params if p.grad is not None]
        if not grads or self.world == 1:
            return
        n = sum(g.numel() for g in grads)
        flat = self._buffer(n, grads[0])
        views, off = [], 0
        for g in grads:
            views.append(flat[off: off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(views, grads)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_(grads, views)

    def mean_scalars(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world


def shard_batch(n_items: int, rank: int, world: int) -> slice:
    """Contiguous, equal shard of a global batch (remainder items go to the lowest ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return slice(start, start + base + (1 if rank < rem else 0))
