"""Adam for modules made of MANY small tensors (CPC: 256 separate ``Wk[i]`` Linear layers + a GRU = 516 tensors).

``torch.optim.Adam(capturable=True)`` keeps one 0-dim device ``step`` tensor PER parameter; its bias-correction
arithmetic runs as foreach ops over lists of 0-dim tensors, which take the per-tensor slow path: ≈4 000 tiny kernels
(fills, adds, divisions) per optimiser step for this module — a tenth of the whole train step.  All parameters of a
group advance together, so ONE shared device counter is enough: the bias corrections become a handful of scalar ops
and the update itself a dozen multi-tensor kernels.  The update formula is torch's (capturable branch):

    m ← β₁m + (1−β₁)g;  v ← β₂v + (1−β₂)g²;  p ← p − (lr / (1−β₁ᵗ)) · m / (√v / √(1−β₂ᵗ) + ε)

Graph-capture safe (the counter lives on the device, no host reads).
"""
from __future__ import annotations

from typing import Iterable

import torch


class SharedStepAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        for group in self.param_groups:
            ps = group["params"]
            dev = ps[0].device
            group["step"] = torch.zeros((), dtype=torch.float32, device=dev)     # shared by every tensor of the group
            group["betas_dev"] = (torch.tensor(group["betas"][0], dtype=torch.float32, device=dev),
                                  torch.tensor(group["betas"][1], dtype=torch.float32, device=dev))   # no H2D inside step()
            for p in ps:
                self.state[p]["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                self.state[p]["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self.state[ps[0]]["step"] = group["step"]                             # visible to state snapshots

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            grads = [p.grad for p in ps]
            m = [self.state[p]["exp_avg"] for p in ps]
            v = [self.state[p]["exp_avg_sq"] for p in ps]
            b1, b2 = group["betas"]
            t = group["step"]
            t += 1
            b1_t, b2_t = group["betas_dev"]
            bc1 = 1.0 - torch.pow(b1_t, t)
            bc2_sqrt = torch.sqrt(1.0 - torch.pow(b2_t, t))
            torch._foreach_mul_(m, b1)
            torch._foreach_add_(m, grads, alpha=1.0 - b1)
            torch._foreach_mul_(v, b2)
            torch._foreach_addcmul_(v, grads, grads, value=1.0 - b2)
            denom = torch._foreach_sqrt(v)
            torch._foreach_div_(denom, bc2_sqrt)
            torch._foreach_add_(denom, group["eps"])
            upd = torch._foreach_div(m, denom)
            torch._foreach_mul_(upd, group["lr"] / bc1)
            torch._foreach_sub_(ps, upd)
        return None
