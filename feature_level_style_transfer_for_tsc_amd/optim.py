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

import ctypes
from typing import Iterable, List, Sequence

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
            if ps[0].is_cuda:
                # one single-pass multi-tensor launch per 64 tensors (csrc/optim.hip) instead of eleven foreach passes
                from . import _lib
                lib = _lib.load()
                t = group["step"]
                t += 1
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
                m = [self.state[p]["exp_avg"] for p in ps]
                v = [self.state[p]["exp_avg_sq"] for p in ps]
                b1, b2 = group["betas"]
                _lib.check(lib.fst_adam_multi(_ptr_array(ps), _ptr_array(grads), _ptr_array(m), _ptr_array(v),
                                              _i64_array([p.numel() for p in ps]), len(ps), t.data_ptr(), group["lr"], b1, b2,
                                              group["eps"], _lib.stream_ptr()), "fst_adam_multi")
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


def _ptr_array(tensors: Sequence[torch.Tensor]):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _i64_array(values: Sequence[int]):
    return (ctypes.c_int64 * len(values))(*values)


class FusedRMSprop(torch.optim.Optimizer):
    """``torch.optim.RMSprop(params, lr)`` with its defaults (alpha 0.99, eps 1e-8, not centered, no momentum, no weight decay —
    what train_and_test.py:97-106 constructs) whose step is ONE pass over every tensor: v ← αv + (1−α)g², p ← p − lr·g/(√v + ε)
    in torch's operation order, up to 64 tensors per launch (csrc/optim.hip).  ``rmsprop_step_many`` steps several of these
    optimisers (one per module, each with its own learning rate) with the same launches: the joint step's ten RMSprops are 4
    launches instead of ~70 foreach launches (1.6 ms of five-pass multi-tensor kernels).  State: ``square_avg`` per parameter,
    created on first use; hipGraph-capture safe once created (the warm-up steps do that)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-2, alpha: float = 0.99, eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        rmsprop_step_many([self])
        return None


@torch.no_grad()
def rmsprop_step_many(opts: Sequence[FusedRMSprop]) -> None:
    ps: List[torch.Tensor] = []
    grads: List[torch.Tensor] = []
    vs: List[torch.Tensor] = []
    lrs: List[float] = []
    alpha = eps = None
    for o in opts:
        for group in o.param_groups:
            if alpha is None:
                alpha, eps = group["alpha"], group["eps"]
            assert (group["alpha"], group["eps"]) == (alpha, eps), "rmsprop_step_many: one (alpha, eps) per call"
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = o.state[p]
                if "square_avg" not in st:
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                ps.append(p)
                grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
                vs.append(st["square_avg"])
                lrs.append(float(group["lr"]))
    if not ps:
        return
    if not ps[0].is_cuda:                                                     # CPU tests of the host logic: torch's own formula
        for p, g, v, lr in zip(ps, grads, vs, lrs):
            v.mul_(alpha).addcmul_(g, g, value=1 - alpha)
            p.addcdiv_(g, v.sqrt().add_(eps), value=-lr)
        return
    from . import _lib
    lib = _lib.load()
    assert all(p.is_contiguous() for p in ps)
    _lib.check(lib.fst_rmsprop_multi(_ptr_array(ps), _ptr_array(grads), _ptr_array(vs), _i64_array([p.numel() for p in ps]),
                                     (ctypes.c_float * len(lrs))(*lrs), len(ps), alpha, eps, _lib.stream_ptr()), "fst_rmsprop_multi")
