"""Drop-in small heads (surface of the reference's widgets.py).  These are a few hundred kFLOP per
sample; they stay stock torch ops (rocBLAS / MIOpen RNN) except the two 1x1 convs over [B, C, L]
tensors, which use the conv engine."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import dist as _dist
from . import ops
from .cdan import calc_coeff as _calc_coeff_cdan, grl_hook


def calc_coeff(iter_num, high=1.0, low=0.0, alpha=2.0, max_iter=50.0):
    return _calc_coeff_cdan(iter_num, high, low, alpha, max_iter)


class _GRLCounter:
    """Gradient-reversal coefficient driven by a per-CALL counter saturating at ``max_iter`` (Q7)."""

    def _init_grl(self):
        self.iter_num = -1
        self.alpha, self.low, self.high, self.max_iter = 100.0, 0.0, 1.0, 20.0

    def _next_coeff(self) -> float:
        if self.training:
            self.iter_num += 1
        if self.iter_num >= self.max_iter:
            self.iter_num = self.max_iter
        return calc_coeff(self.iter_num, self.high, self.low, self.alpha, self.max_iter)


class FeatureDiscriminatorforSource(nn.Module, _GRLCounter):
    """GRL → 4-layer LeakyReLU MLP (widgets.py:15-42)."""

    def __init__(self, length_of_feature):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(length_of_feature, 800), nn.LeakyReLU(0.2), nn.Linear(800, 400),
                                   nn.LeakyReLU(0.2), nn.Linear(400, 50), nn.LeakyReLU(0.2), nn.Linear(50, 1))
        self._init_grl()

    def forward(self, probs):
        coeff = self._next_coeff()
        probs = probs * 1.0
        probs.register_hook(grl_hook(coeff))
        m = self.model
        # Linear + LeakyReLU pairs as one GEMM each (bias and activation in its epilogue)
        h = ops.linear_act(probs, m[0], ops.ACT_LEAKY, m[1].negative_slope)
        h = ops.linear_act(h, m[2], ops.ACT_LEAKY, m[3].negative_slope)
        h = ops.linear_act(h, m[4], ops.ACT_LEAKY, m[5].negative_slope)
        return ops.linear_act(h, m[6])


class ProbTransfer(nn.Module):
    """LSTM over the pooled feature repeated twice; returns h_n (widgets.py:46-55)."""

    def __init__(self, num_of_channels):
        super().__init__()
        self.model = nn.LSTM(input_size=num_of_channels, hidden_size=num_of_channels, batch_first=True)

    def forward(self, features_of_target_before_linear):
        x = features_of_target_before_linear
        m = self.model
        if x.is_cuda and x.dim() == 2 and m.hidden_size <= 256:
            # both steps see the same input: one projection (b_hh is added at every step, also at step 1 where h0 = 0),
            # then the two-step recurrence as one launch
            xproj = torch.addmm(m.bias_ih_l0 + m.bias_hh_l0, x, m.weight_ih_l0.t())
            return ops.LSTM2Fn.apply(xproj, m.weight_hh_l0)
        x = torch.unsqueeze(x, 1)
        _, (h_n, _) = m(torch.cat((x, x), dim=1))
        return torch.squeeze(h_n, dim=0)


def wgan_loss(values_from_target_side, values_from_s2t2s, values_from_source_side):
    return -torch.mean(values_from_target_side) - torch.mean(values_from_s2t2s) + torch.mean(values_from_source_side)


class DimensionUnification(nn.Module):
    """ReLU(Linear over time) → ReLU(1x1 conv over channels) (widgets.py:66-78)."""

    def __init__(self, source_channel, target_channel, source_length, target_length):
        super().__init__()
        self.length_unification = nn.Linear(in_features=source_length, out_features=target_length)
        self.relu1 = nn.ReLU()
        self.channel_unification = nn.Conv1d(in_channels=source_channel, out_channels=target_channel, kernel_size=1)
        self.relu2 = nn.ReLU()
        self.spec = ops.ConvSpec(target_channel, source_channel)

    def forward(self, source_feature):
        if source_feature.is_cuda and source_feature.dtype == torch.float32:
            # both ReLUs in the epilogues of their GEMMs: two launches for the module
            h = ops.linear_act(source_feature, self.length_unification, ops.ACT_RELU)
            return ops.ConvReluFn.apply(self.spec, h, self.channel_unification.weight, self.channel_unification.bias)
        h = self.relu1(self.length_unification(source_feature))
        h = ops.conv1d(self.spec, h.contiguous(), self.channel_unification.weight, self.channel_unification.bias)
        return self.relu2(h)


def init_weights(m):
    """``ad_net.apply(init_weights)`` (widgets.py:84-94 of the reference): Xavier-normal Linear weights, zero biases;
    BatchNorm layers, if a head ever holds one, N(1, 0.02) weights."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.modules.batchnorm._BatchNorm):
        nn.init.normal_(m.weight, 1.0, 0.02)
        nn.init.zeros_(m.bias)


class AdversarialNetworkforCDAN(nn.Module, _GRLCounter):
    """GRL → Linear-ReLU-Dropout ×2 → Linear(1) (widgets.py:95-131); ``coeff`` is read by CDAN()."""

    def __init__(self, in_feature, hidden_size):
        super().__init__()
        self.ad_layer1 = nn.Linear(in_feature, hidden_size)
        self.ad_layer2 = nn.Linear(hidden_size, hidden_size)
        self.ad_layer3 = nn.Linear(hidden_size, 1)
        self.relu1, self.relu2 = nn.ReLU(), nn.ReLU()
        self.dropout1, self.dropout2 = nn.Dropout(0.2), nn.Dropout(0.2)
        self.apply(init_weights)
        self._init_grl()
        self.coeff = float(0.001)

    def forward(self, x):
        coeff = self._next_coeff()
        self.coeff = coeff
        x = x * 1.0
        x.register_hook(grl_hook(coeff))
        x = self.dropout1(ops.linear_act(x, self.ad_layer1, ops.ACT_RELU))
        x = self.dropout2(ops.linear_act(x, self.ad_layer2, ops.ACT_RELU))
        return ops.linear_act(x, self.ad_layer3)


class NoiseTransfer(nn.Module):
    """Latent mean-shift "style transfer" (widgets.py:136-167) with the reference's stateful, detached
    running sums (Q5: they are not averages — ``avg += (B/N_seen)·mean(batch)``)."""

    def __init__(self, noise_channel, length_of_noise, with_nvidia=True):
        super().__init__()
        self.apply_learnable_weight = nn.Conv1d(noise_channel, noise_channel, 1)
        self.activation_selu = nn.SELU()
        self.target_avg = torch.zeros([noise_channel, length_of_noise]).float()
        self.source_avg = torch.zeros([noise_channel, length_of_noise]).float()
        self.time = 0
        self.cal_num_target = 0
        self.cal_num_source = 0

    def _apply(self, fn, *a, **k):
        self.target_avg, self.source_avg = fn(self.target_avg), fn(self.source_avg)
        return super()._apply(fn, *a, **k)

    def advance(self, batch_target: int, batch_source: int):
        """Host-side bookkeeping of one call (:151-161): returns the two accumulation ratios
        (1 on the first call, B/N_seen afterwards — Q5) and bumps the counters."""
        self.time += 1
        if self.time == 1:
            ratios = (1.0, 1.0)
        else:
            ratios = (batch_target / self.cal_num_target, batch_source / self.cal_num_source)
        self.cal_num_target += batch_target
        self.cal_num_source += batch_source
        return ratios

    def forward(self, target_noise_batch, source_noise_batch, ratios=None):
        """``ratios``: optional pair of 0-d DEVICE tensors holding this call's accumulation ratios; a captured
        hipGraph passes static buffers that the host refreshes (via ``advance``) before every replay."""
        if ratios is None:
            ratios = self.advance(target_noise_batch.size(0), source_noise_batch.size(0))
        if (source_noise_batch.is_cuda and not _dist.global_batch_active() and target_noise_batch.shape == source_noise_batch.shape
                and (source_noise_batch.size(1) * source_noise_batch.size(2)) % 4 == 0 and source_noise_batch.dtype == torch.float32):
            # three launches (csrc/widgets.hip); the running sums are updated in place by the kernel
            return ops.NoiseTransferFn.apply(target_noise_batch, source_noise_batch, self.apply_learnable_weight.weight,
                                             self.apply_learnable_weight.bias, self.target_avg, self.source_avg, ratios[0], ratios[1])
        # (global-batch data parallelism: the batch means run over every rank's samples)
        new_target = self.target_avg + ratios[0] * _dist.mean_over_ranks(torch.mean(target_noise_batch, dim=0))
        new_source = self.source_avg + ratios[1] * _dist.mean_over_ranks(torch.mean(source_noise_batch, dim=0))
        general_distance = new_target - new_source
        learned = self.activation_selu(self.apply_learnable_weight(general_distance))   # unbatched [C, L] conv
        self.target_avg.copy_(new_target.detach())                # state is kept detached, in place
        self.source_avg.copy_(new_source.detach())
        return learned + source_noise_batch
