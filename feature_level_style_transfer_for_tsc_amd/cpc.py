"""Drop-in CPC self-supervised loss (surface of ``CPC`` in the reference's Comparison/SLARDA/train.py:41-76).

The reference loops T = L/2 times in Python, building T separate B×B Gram matrices on the CPU.  Here the
T linear predictors are ONE batched GEMM and all T cross-Grams + log-softmax + diagonal pick are one
fused HIP kernel reading the encodings in place from the [B, C, L] feature tensor (ops.CPCNceFn).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

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

    def forward(self, features: torch.Tensor, t_samples: Optional[int] = None) -> torch.Tensor:
        """``t_samples`` pins the random start (quirk Q6); by default it is drawn from the global CPU RNG
        exactly as the reference does (:58)."""
        if t_samples is None:
            t_samples = int(torch.randint(self.timestep // 2, size=(1,)).long())
        B, C, L = features.shape
        T = self.timestep
        z = features.transpose(1, 2)
        output, _ = self.gru(z[:, : t_samples + 1, :].contiguous())
        c_t = output[:, t_samples, :].reshape(B, self.hidden_dim)
        W = torch.stack([l.weight for l in self.Wk])                          # [T, C, H]
        b = torch.stack([l.bias for l in self.Wk])                            # [T, C]
        pred = torch.baddbmm(b.unsqueeze(1), c_t.unsqueeze(0).expand(T, B, -1), W.transpose(1, 2))   # [T, B, C]
        return ops.CPCNceFn.apply(features, pred, t_samples + 1, T)
