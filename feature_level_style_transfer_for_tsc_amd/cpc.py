"""Drop-in CPC self-supervised loss (surface of ``CPC`` in the reference's Comparison/SLARDA/train.py:41-76).

The reference loops T = L/2 times in Python, building T separate B×B Gram matrices on the CPU.  Here the
T linear predictors are ONE batched GEMM and all T cross-Grams + log-softmax + diagonal pick are one
fused HIP kernel reading the encodings in place from the [B, C, L] feature tensor (ops.CPCNceFn).
"""
from __future__ import annotations

from typing import Optional

import os

import torch
import torch.nn as nn

from . import dist as _dist
from . import ops


class CPC(nn.Module):
    def __init__(self, num_channels, gru_hidden_dim, timestep):
        super().__init__()
        self.num_channels = num_channels
        self.hidden_dim = gru_hidden_dim
        self.gru = nn.GRU(num_channels, self.hidden_dim, num_layers=1, bidirectional=False, batch_first=True)
        self.timestep = timestep
        self.Wk = nn.ModuleList([nn.Linear(self.hidden_dim, num_channels) for _ in range(self.timestep)])
        self.lsoftmax = nn.LogSoftmax(dim=-1)
        self._stack_cache = None

    def _stacked(self):
        """The T predictors as one [T, C, H] / [T, C] pair.  Inside ``shared_stack()`` the pair is built once and
        reused by every call (the joint step runs CPC twice): the gradient then reaches the 2·T separate parameters
        through ONE unbind instead of being accumulated tensor by tensor per call."""
        if self._stack_cache is not None and self._stack_cache[0] is not None:
            return self._stack_cache
        pair = (torch.stack([l.weight for l in self.Wk]), torch.stack([l.bias for l in self.Wk]))
        if self._stack_cache is not None:
            self._stack_cache = pair
        return pair

    def shared_stack(self):
        mod = self

        class _Ctx:
            def __enter__(self_):
                self_.prev, mod._stack_cache = mod._stack_cache, (None, None)

            def __exit__(self_, *exc):
                mod._stack_cache = self_.prev
                return False
        return _Ctx()

    def forward(self, features: torch.Tensor, t_samples=None) -> torch.Tensor:
        """``t_samples`` pins the random start (quirk Q6); by default it is drawn from the global CPU RNG
        exactly as the reference does (:58).  It may also be a 0-d int32 DEVICE tensor: then every shape is
        static (the GRU runs all T/2 steps and its t-th output is gathered), which lets a captured hipGraph
        replay the step with a new start index each time; results are identical because the GRU is causal."""
        if t_samples is None:
            t_samples = int(torch.randint(self.timestep // 2, size=(1,)).long())
        B, C, L = features.shape
        T = self.timestep
        z = features.transpose(1, 2)
        dev_t = isinstance(t_samples, torch.Tensor)
        S = max(1, T // 2) if dev_t else t_samples + 1          # device index: static shapes, the recurrence stops at t
        if self.hidden_dim == 64 and features.is_cuda:
            # the input projections of all S steps as one GEMM, then the recurrence as ONE persistent launch
            # (ops.GRULastFn) — nn.GRU's parameters are used as they are (torch's r | z | n gate order)
            if ops.MATH == "bf16x3" and os.environ.get("FST_CPC_GEMM", "1") != "0":      # (0: diagnostics, the library GEMM)
                xproj = ops.LinearActFn.apply(z[:, :S, :], self.gru.weight_ih_l0, self.gru.bias_ih_l0, ops.ACT_NONE, 0.0)
            else:
                xproj = torch.matmul(z[:, :S, :], self.gru.weight_ih_l0.t()) + self.gru.bias_ih_l0
            c_t = ops.GRULastFn.apply(xproj, self.gru.weight_hh_l0, self.gru.bias_hh_l0, t_samples)
        elif dev_t:
            output, _ = self.gru(z[:, :S, :].contiguous())
            idx = t_samples.long().view(1, 1, 1).expand(B, 1, self.hidden_dim)
            c_t = output.gather(1, idx).reshape(B, self.hidden_dim)
        else:
            output, _ = self.gru(z[:, :S, :].contiguous())
            c_t = output[:, t_samples, :].reshape(B, self.hidden_dim)
        t0 = (t_samples + 1).to(torch.int32) if dev_t else t_samples + 1
        W, b = self._stacked()
        # predᵀ[i] = W_i·c_tᵀ + b_i: with W as the LEFT operand its gradient dpredᵀ·c_t comes out contiguous [T, C, H], so the
        # stack's backward hands every Wk[i] a slice autograd can keep as it is (as the right operand the gradient is a
        # transposed view and each of the T = L/2 predictors costs a strided clone: 256 launches per CPC call)
        pred = torch.baddbmm(b.unsqueeze(2), W, c_t.t().unsqueeze(0).expand(T, -1, -1)).transpose(1, 2)   # [T, B, C]
        # global-batch data parallelism: the negatives of a row are the predictions of EVERY rank's samples
        return ops.CPCNceFn.apply(features, _dist.gather_cat(pred, 1), t0, T, _dist.rank() * B)
