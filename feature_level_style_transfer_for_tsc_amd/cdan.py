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
    def reverse(grad):
        return -coeff * grad.clone()
    return reverse


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


def CDAN(input_target, input_g_from_source, prob_target, prob_g_from_source, ad_net, random_layer=None):
    """Entropy-weighted Wasserstein-style CDAN distance (C_DAN.py:49-82), including quirk Q4: the
    ``view(-1, 1)`` results are discarded, so ``[B] * [B, 1]`` broadcasts to ``[B, B]``."""
    input_target = torch.flatten(input_target, 1)
    input_g_from_source = torch.flatten(input_g_from_source, 1)
    prob_target = torch.nn.functional.softmax(prob_target, dim=1)
    prob_g_from_source = torch.nn.functional.softmax(prob_g_from_source, dim=1)
    if random_layer is None:
        fusion_target = torch.bmm(prob_target.unsqueeze(2), input_target.unsqueeze(1))
        target_out = ad_net(fusion_target.view(-1, input_target.size(1) * prob_target.size(1)))
        fusion_source = torch.bmm(prob_g_from_source.unsqueeze(2), input_g_from_source.unsqueeze(1))
        g_source_out = ad_net(fusion_source.view(-1, input_g_from_source.size(1) * prob_g_from_source.size(1)))
    else:
        target_out = ad_net(random_layer.forward([input_target, prob_target]))
        g_source_out = ad_net(random_layer.forward([input_g_from_source, prob_g_from_source]))
    entropy_target = Entropy(prob_target)
    entropy_g_from_source = Entropy(prob_g_from_source)
    coeff = ad_net.coeff
    entropy_target.register_hook(grl_hook(coeff))
    entropy_g_from_source.register_hook(grl_hook(coeff))
    weight_target = 1.0 + torch.exp(-entropy_target)
    weight_g_from_source = 1.0 + torch.exp(-entropy_g_from_source)
    if _dist.global_batch_active():
        return _q4_sum_global(weight_target, target_out) - _q4_sum_global(weight_g_from_source, g_source_out)
    weight_target = weight_target / torch.sum(weight_target).detach()
    weight_g_from_source = weight_g_from_source / torch.sum(weight_g_from_source).detach()
    distance_target = torch.sum(weight_target * target_out)                 # [B]·[B,1] → [B,B] (Q4)
    distance_g_from_source = torch.sum(weight_g_from_source * g_source_out)
    return distance_target - distance_g_from_source


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
