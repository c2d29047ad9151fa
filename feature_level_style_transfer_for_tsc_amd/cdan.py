"""Drop-in conditional-adversarial pieces (surface of the reference's C_DAN.py).

``RandomLayer``'s [B, C·L] × [C·L, 1024] product — the one large GEMM of the head, HBM-bound on the
fixed 105 MB matrix — runs on the MFMA conv engine with a K split; the class-probability product and
the Hadamard stay tiny torch ops.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import dist as _dist
from . import ops


def calc_coeff(iter_num, high=1.0, low=0.0, alpha=100.0, max_iter=50.0):
    return float(2.0 * (high - low) / (1.0 + np.exp(-alpha * iter_num / max_iter)) - (high - low) + low)


def grl_hook(coeff):
    """Tensor hook form of the gradient reversal (the small heads in widgets.py register it on their inputs)."""
    return lambda grad: -coeff * grad


class RandomLayer(nn.Module):
    """Random multilinear map (C_DAN.py:11-25).  The matrices are fixed Gaussians, not Parameters and
    not in the state_dict (as in the reference); ``.cuda()``/``.to()`` move them."""

    def __init__(self, input_dim_list=[], output_dim=1024, with_nvidia=True):
        super().__init__()
        self.input_num = len(input_dim_list)
        self.output_dim = output_dim
        self.random_matrix = [torch.randn(input_dim_list[i], output_dim) for i in range(self.input_num)]
        self._transposed = {}

    def _apply(self, fn, *a, **k):
        self.random_matrix = [fn(m) for m in self.random_matrix]
        self._transposed = {}
        return super()._apply(fn, *a, **k)

    def _rt(self, i: int) -> torch.Tensor:
        if i not in self._transposed:
            self._transposed[i] = self.random_matrix[i].t().contiguous()
        return self._transposed[i]

    def forward(self, input_list):
        if self.input_num == 2:
            x, pr = input_list
            R0, R1 = self.random_matrix
            if (x.dim() == 2 and R0.shape[0] >= 256 and not R1.requires_grad
                    and ops.nt_gemm_ok(x.shape[0], R0.shape[1], R0.shape[0], x.contiguous(), R0, self._rt(0))
                    and ops.nt_gemm_ok(x.shape[0], R0.shape[0], R0.shape[1], R0)):
                # one GEMM with the class-side product, the 1/√O scale and the Hadamard product in its epilogue
                return ops.RandomLayerFn.apply(x, pr, R0, self._rt(0), R1.contiguous(), 1.0 / math.pow(float(self.output_dim), 0.5))
        outs = []
        for i in range(self.input_num):
            R = self.random_matrix[i]
            if R.shape[0] >= 256:                                           # the feature-side GEMM
                outs.append(ops.FixedMatmulFn.apply(input_list[i], R, self._rt(i)))
            else:                                                           # [B, n_class] × [n_class, O]
                outs.append(torch.mm(input_list[i], R))
        result = outs[0] / math.pow(float(self.output_dim), 1.0 / len(outs))
        for single in outs[1:]:
            result = torch.mul(result, single)
        return result


def Entropy(input_):
    epsilon = 1e-5
    return torch.sum(-input_ * torch.log(input_ + epsilon), dim=1)


class _ReverseGrad(torch.autograd.Function):
    """Identity forward, cotangent × (−coeff) backward — the gradient-reversal the reference installs with a tensor hook."""

    @staticmethod
    def forward(ctx, x, coeff: float):
        ctx.coeff = coeff
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.coeff * g, None


def _critic_side(features, logits, ad_net, random_layer):
    """One domain's half of the loss: (class probabilities, critic outputs [B, 1]).  The critic sees the random
    multilinear map of (flattened features, class probabilities) — or, without a random layer, their full outer
    product p ⊗ f (C_DAN.py:57-61)."""
    f = torch.flatten(features, 1)
    prob = torch.softmax(logits, dim=1)
    if random_layer is not None:
        joint = random_layer.forward([f, prob])
    else:
        joint = (prob.unsqueeze(2) * f.unsqueeze(1)).reshape(f.size(0), -1)      # [B, n_class·D], class-major like the bmm
    return prob, ad_net(joint)


def _entropy_weights(prob, coeff: float):
    """w_b = 1 + e^{−H(p_b)}, the entropy's gradient reversed and scaled by ``coeff`` (C_DAN.py:66-72)."""
    return 1.0 + torch.exp(-_ReverseGrad.apply(Entropy(prob), coeff))


def _q4_sum(w: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """Quirk Q4 (C_DAN.py:74-80): the reference normalises w by its detached sum S, discards the ``view(-1, 1)`` and
    multiplies [B]·[B, 1], which broadcasts to [B, B]; the sum of that matrix is (Σ_b w_b / S)·(Σ_b out_b) — written as
    that product here (same value, same gradients: ∂/∂w_b = Σout/S, ∂/∂out_b = Σw/S = 1)."""
    if _dist.global_batch_active():
        return _q4_sum_global(w, out)
    return (torch.sum(w) / torch.sum(w).detach()) * torch.sum(out)


def CDAN(input_target, input_g_from_source, prob_target, prob_g_from_source, ad_net, random_layer=None):
    """Entropy-weighted Wasserstein-style CDAN distance (C_DAN.py:49-82): target side minus transferred-source side,
    each side summed as quirk Q4 prescribes.  ``prob_*`` are logits (softmax is taken here, as in the reference)."""
    p_t, out_t = _critic_side(input_target, prob_target, ad_net, random_layer)
    p_g, out_g = _critic_side(input_g_from_source, prob_g_from_source, ad_net, random_layer)
    # the critic's GRL coefficient advances with every forward call; the reference reads it once, AFTER both calls, and
    # reverses both entropy gradients with that value (C_DAN.py:62-72) — on the first batch 0.987, not the 0 of call one
    coeff = ad_net.coeff
    w_t, w_g = _entropy_weights(p_t, coeff), _entropy_weights(p_g, coeff)
    return _q4_sum(w_t, out_t) - _q4_sum(w_g, out_g)


def _q4_sum_global(w: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """Q4 over the samples of every rank: the [B]·[B,1] broadcast sums to (Σ_b w_b / S)·(Σ_b out_b) with
    S = Σ_b w_b detached — a product of two GLOBAL batch sums, W_g·O_g / S_g.  Each rank returns the global value and
    carries the gradient of its own samples, scaled by the world size because the bucket later averages parameter
    gradients while this term is a sum over the batch, not a mean:  ∂/∂w_b = O_g/S_g,  ∂/∂out_b = W_g/S_g = 1."""
    n = _dist.world()
    W_r, O_r = torch.sum(w), torch.sum(out)
    tot = torch.stack([W_r.detach(), O_r.detach()])
    _dist.sum_over_ranks_(tot)
    W_g, O_g = tot[0], tot[1]
    carrier = n * ((W_r / W_g) * O_g + O_r)
    return carrier - carrier.detach() + O_g
