"""Drop-in simplified WaveGlow (constructor / forward / infer / state_dict surface of the reference's
Simplified_NF_WaveGlow.py) on the HIP conv engine.

Per flow: invertible 1x1 conv (MFMA GEMM over channels) → WN on the first half of the channels (one
autograd node, ops.WNFn) → affine coupling (one pointwise kernel).  ``weight_norm``-wrapped ``nn.Conv1d``
objects are parameter containers only (old-style ``weight_g`` / ``weight_v`` names, as in the reference);
the effective weights g·v/‖v‖ are folded once per call and handed to the kernels.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import ops


class Invertible1x1Conv(nn.Module):
    """z = W·x per timestep, log_det = B·L·logdet(W); reverse uses an inverse computed ONCE and cached
    without gradient (quirk Q2, Simplified_NF_WaveGlow.py:24-42)."""

    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv1d(c, c, kernel_size=1, stride=1, padding=0, bias=False)
        W = torch.linalg.qr(torch.FloatTensor(c, c).normal_())[0]          # random orthonormal init (:17)
        if torch.det(W) < 0:
            W[:, 0] = -1 * W[:, 0]
        self.conv.weight.data = W.contiguous().view(c, c, 1)
        self.spec = ops.ConvSpec(c, c)
        self._logdet_cache = None

    def forward(self, z: torch.Tensor, reverse: bool = False):
        batch_size, _, n_of_groups = z.size()
        if reverse:
            if not hasattr(self, "W_inverse"):
                self.W_inverse = self.conv.weight.detach().squeeze().float().inverse()[..., None].contiguous()
            return ops.conv1d(self.spec, z.contiguous(), self.W_inverse, None)
        W = self.conv.weight
        # inside WaveGlow.shared_fold() (one train step: W is the same for both forward passes) logdet(W) is taken once and its
        # autograd node shared
        cache = self._logdet_cache
        if cache is not None and cache[0] is not None:
            logdet = cache[0]
            if len(cache) > 1 and cache[1] is not None:                 # taken ahead of time on another stream (prefetch_logdets)
                torch.cuda.current_stream().wait_event(cache[1])
                logdet.record_stream(torch.cuda.current_stream())
                cache[1] = None
        else:
            logdet = ops.logdet(W.squeeze())
            if cache is not None:
                cache[0] = logdet
        log_det_W = batch_size * n_of_groups * logdet
        return ops.conv1d(self.spec, z.contiguous(), W, None), log_det_W


def _wn_conv(cin: int, cout: int, k: int, **kw) -> nn.Conv1d:
    return nn.utils.weight_norm(nn.Conv1d(cin, cout, k, **kw), name="weight")


def _folded(conv: nn.Conv1d) -> torch.Tensor:
    return torch._weight_norm(conv.weight_v, conv.weight_g, 0)


class WN(nn.Module):
    """Gated dilated-conv stack conditioned on its own input (Simplified_NF_WaveGlow.py:55-123)."""

    def __init__(self, n_in_channels: int, n_layers: int, n_channels: int, kernel_size: int):
        super().__init__()
        self.n_layers, self.n_channels = n_layers, n_channels
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        self.start = _wn_conv(n_in_channels, n_channels, 1)
        end = nn.Conv1d(n_channels, 2 * n_in_channels, 1)
        end.weight.data.zero_()                                            # coupling starts as identity (:73-77)
        end.bias.data.zero_()
        self.end = end
        self.cond_layer = _wn_conv(n_in_channels, 2 * n_channels * n_layers, 1)
        for i in range(n_layers):
            d = 2 ** i
            self.in_layers.append(_wn_conv(n_channels, 2 * n_channels, kernel_size, dilation=d,
                                           padding=int((kernel_size * d - d) / 2)))
            self.res_skip_layers.append(_wn_conv(n_channels, 2 * n_channels if i < n_layers - 1 else n_channels, 1))
        self.specs = ops.WNSpecs(n_in_channels, n_channels, n_layers, kernel_size)
        self._fold_cache = None
        self._fold_plan = None

    def folded_weights(self) -> torch.Tensor:
        """Effective weights g·v/‖v‖ as ONE flat tensor in ``WNSpecs.shapes`` order.  Inside ``WaveGlow.shared_fold()`` (one
        train step: the weights are the same for the two forward passes and ``infer``) the fold is done once and its
        autograd graph is shared: the three applications' gradients are summed on the flat tensor (two adds per WN)."""
        return self._folded()[0]

    def _folded(self):
        """(flat weights, the WNGradPool their gradient is joined through — None outside ``shared_fold`` and on the CPU)."""
        if self._fold_cache is not None and self._fold_cache[0] is not None:
            return self._fold_cache[0]
        pool = None
        if self.start.weight_v.is_cuda:
            flat = ops.WNFoldFn.apply(self._plan(), *self._fold_inputs())        # one launch (18 weight-norm launches + a cat as torch ops)
            if self._fold_cache is not None and torch.is_grad_enabled() and flat.requires_grad:
                # the applications made in this scope leave their weight-gradient operands in the pool; the join node sums them
                # with one launch per layer (the slab reduction of the time-as-k kernels is paid once, not per application)
                pool = ops.WNGradPool()
                flat = ops.WGradJoinFn.apply(pool, self.specs, flat)
        else:
            flat = self.specs.flatten(self._fold())
        if self._fold_cache is not None:
            self._fold_cache[0] = (flat, pool)
        return flat, pool

    def _folded_for(self, inverse: bool):
        flat, pool = self._folded()
        return flat, inverse, pool

    def _plan(self) -> "ops.WNFoldPlan":
        if self._fold_plan is None:
            nl = self.n_layers
            normed = [True, False, True, False, False, False] + [True] * nl + [False] * nl + [True] * nl + [False] * nl
            self._fold_plan = ops.WNFoldPlan(self.specs, normed)
        return self._fold_plan

    def _fold_inputs(self) -> List[torch.Tensor]:
        """Parameter tensors in ``WNSpecs.shapes`` order, (v, g) for a weight-normed conv's weight."""
        t = [self.start.weight_v, self.start.weight_g, self.start.bias, self.cond_layer.weight_v, self.cond_layer.weight_g,
             self.cond_layer.bias, self.end.weight, self.end.bias]
        for l in self.in_layers:
            t += [l.weight_v, l.weight_g]
        t += [l.bias for l in self.in_layers]
        for l in self.res_skip_layers:
            t += [l.weight_v, l.weight_g]
        t += [l.bias for l in self.res_skip_layers]
        return t

    def _fold(self) -> List[torch.Tensor]:
        w = [_folded(self.start), self.start.bias, _folded(self.cond_layer), self.cond_layer.bias,
             self.end.weight, self.end.bias]
        w += [_folded(l) for l in self.in_layers] + [l.bias for l in self.in_layers]
        w += [_folded(l) for l in self.res_skip_layers] + [l.bias for l in self.res_skip_layers]
        return w

    def forward(self, forward_input: torch.Tensor) -> torch.Tensor:
        return ops.WNFn.apply(self.specs, forward_input, self.folded_weights())


class WaveGlow(nn.Module):
    """``n_flows`` × {invertible 1x1, WN, affine coupling} (Simplified_NF_WaveGlow.py:125-203)."""

    def __init__(self, n_flows: int, n_group: int, n_channels_for_WN: int):
        super().__init__()
        assert n_group % 2 == 0
        self.n_flows, self.n_group = n_flows, n_group
        self.WN = nn.ModuleList()
        self.convinv = nn.ModuleList()
        for _ in range(n_flows):
            self.convinv.append(Invertible1x1Conv(n_group))
            self.WN.append(WN(n_group // 2, 8, n_channels_for_WN, 3))

    class _SharedFold:
        def __init__(self, wg):
            self.wg = wg

        def __enter__(self):
            for wn in self.wg.WN:
                wn._fold_cache = [None]
            for c in self.wg.convinv:
                c._logdet_cache = [None]
            return self

        def __exit__(self, *exc):
            for wn in self.wg.WN:
                wn._fold_cache = None
            for c in self.wg.convinv:
                c._logdet_cache = None
            return False

    def shared_fold(self):
        """Context manager: fold the weight-norm parameters once for every pass made inside it."""
        return WaveGlow._SharedFold(self)

    def prefetch_logdets(self, stream: "torch.cuda.Stream") -> None:
        """Inside ``shared_fold()``: take log|det W| of every flow's 1x1 weights now, on ``stream`` — each is a single-workgroup
        Gauss-Jordan elimination (0.19 ms at 50 channels) that depends on nothing but the weights, so it need not sit in the
        chain of the first forward pass; the pass waits for the event where it first uses the value."""
        if not self.convinv[0].conv.weight.is_cuda:
            return
        with torch.cuda.stream(stream):
            for c in self.convinv:
                if c._logdet_cache is not None and c._logdet_cache[0] is None:
                    c._logdet_cache[0] = ops.logdet(c.conv.weight.squeeze())
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    c._logdet_cache.append(ev)

    def forward(self, forward_input: torch.Tensor):
        audio = forward_input
        log_s_list, log_det_W_list = [], []
        n_half = self.n_group // 2
        for k in range(self.n_flows):
            audio, log_det_W = self.convinv[k](audio)
            log_det_W_list.append(log_det_W)
            # WN on the first half of the channels + affine coupling as ONE autograd node (ops.FlowFn); Σ log_s of this flow (and
            # Σ z² after the last one) are reduced inside the coupling kernel: WaveGlowLoss picks them up from the attributes below
            # instead of re-reading the tensors (the returned tensors themselves are the reference's)
            audio, output, s_ls, s_sq = ops.FlowFn.apply(self.WN[k].specs, audio, *self.WN[k]._folded_for(False))
            log_s = output[:, n_half:, :]
            log_s._fst_sum = s_ls
            audio._fst_sq_sum = s_sq
            log_s_list.append(log_s)
        return audio, log_s_list, log_det_W_list

    def infer(self, audio: torch.Tensor, sigma: float = 1.0) -> torch.Tensor:
        n_half = self.n_group // 2
        for k in reversed(range(self.n_flows)):
            audio = ops.FlowFn.apply(self.WN[k].specs, audio, *self.WN[k]._folded_for(True))[0]
            audio = self.convinv[k](audio, reverse=True)
        return audio


class WaveGlowLoss(nn.Module):
    """(Σz²/2σ² − Σlog_s − Σlog_det_W) / (B·C·L) (Simplified_NF_WaveGlow.py:223-241)."""

    def __init__(self, sigma: float = 1.0):
        super().__init__()
        self.sigma = sigma

    def forward(self, model_output):
        z, log_s_list, log_det_W_list = model_output
        # sums already reduced by the coupling kernels (WaveGlow.forward) when present; any other tensors are summed here
        total = lambda t, attr: getattr(t, attr) if hasattr(t, attr) else None
        sums_ls = [total(t, "_fst_sum") for t in log_s_list]
        log_s_total = sums_ls[0] if sums_ls[0] is not None else log_s_list[0].sum()
        log_det_W_total = log_det_W_list[0]
        for log_s, s_ls, log_det in zip(log_s_list[1:], sums_ls[1:], log_det_W_list[1:]):
            log_s_total = log_s_total + (s_ls if s_ls is not None else log_s.sum())
            log_det_W_total = log_det_W_total + log_det
        sq = total(z, "_fst_sq_sum")
        loss = (sq if sq is not None else torch.sum(z * z)) / (2 * self.sigma * self.sigma) - log_s_total - log_det_W_total
        return loss / (z.size(0) * z.size(1) * z.size(2))
