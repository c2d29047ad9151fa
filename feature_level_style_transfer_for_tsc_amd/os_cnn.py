"""Drop-in OS-CNN modules (constructor / forward / state_dict surface of the reference's
OS_CNN/OS_CNN.py) running on the HIP conv engine.

An omni-scale layer = every prime-kernel branch of the layer packed into ONE kernel launch: the input
window is staged once in LDS and each 32-channel block of branches multiplies only its own live taps
(the reference convolves a dense Kmax kernel whose masked taps are zeros, OS_CNN.py:67-71).
``nn.Conv1d`` / ``nn.BatchNorm1d`` objects are used purely as parameter containers so that parameter
names, shapes, init laws and RNG consumption match the reference; their ``forward`` is never called.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from .structure import (LayerSpec, calculate_mask_index, layer_parameter_list_input_change, out_channels,
                        row_live_ranges)

__all__ = ["build_layer_with_layer_parameter", "OS_CNN", "OS_block", "SampaddingConv1D_BN", "Res_OS_layer",
           "OS_CNN_res", "layer_parameter_list_input_change", "calculate_mask_index"]


class build_layer_with_layer_parameter(nn.Module):
    """One omni-scale layer: packed multi-kernel conv → BatchNorm1d → optional ReLU (OS_CNN.py:46-77).

    ``dense_weight_grad`` (default True) reproduces quirk Q1: the reference's weight gradient is dense
    over Kmax (masked taps get gradients that GradNorm's norms include, train_and_test.py:685-690).
    """

    def __init__(self, layer_parameters: LayerSpec, relu_or_not_at_last_layer: bool = True, with_nvidia: bool = True,
                 dense_weight_grad: bool = True):
        super().__init__()
        self.relu_or_not_at_last_layer = relu_or_not_at_last_layer
        self.layer_parameters = [tuple(t) for t in layer_parameters]
        cin = self.layer_parameters[0][0]
        kmax = self.layer_parameters[-1][2]
        cout = out_channels(self.layer_parameters)
        # Per-branch default Conv1d init (fan-in = cin·p), drawn in branch order, then the container conv
        # — the same RNG consumption as the reference (Q9, OS_CNN.py:28-35,61-63).
        weight = torch.zeros(cout, cin, kmax)
        bias = torch.zeros(cout)
        row = 0
        for (ci, width, k) in self.layer_parameters:
            branch = nn.Conv1d(ci, width, k)
            lo, hi = calculate_mask_index(k, kmax)
            weight[row: row + width, :, lo:hi] = branch.weight.detach()
            bias[row: row + width] = branch.bias.detach()
            row += width
        self.conv1d = nn.Conv1d(cin, cout, kmax)
        self.conv1d.weight = nn.Parameter(weight)
        self.conv1d.bias = nn.Parameter(bias)
        self.bn = nn.BatchNorm1d(cout)
        live = row_live_ranges(self.layer_parameters)
        self._live_lo_host = torch.tensor([r[0] for r in live], dtype=torch.int32)
        self._live_hi_host = torch.tensor([r[1] for r in live], dtype=torch.int32)
        self._live_dev = {}
        self.spec = ops.ConvSpec(cout, cin, kmax, 1, int((kmax - 1) / 2), row_live=live, dense_dw=dense_weight_grad)

    @property
    def weight_mask(self) -> torch.Tensor:
        """The 0/1 mask the reference keeps as a plain attribute (not in the state_dict)."""
        kmax = self.spec.ntaps
        k = torch.arange(kmax).view(1, 1, kmax)
        lo, hi = self._live_lo_host.view(-1, 1, 1), self._live_hi_host.view(-1, 1, 1)
        return ((k >= lo) & (k < hi)).float().expand(-1, self.spec.C0, -1).contiguous()

    def _live(self, device):
        key = str(device)
        if key not in self._live_dev:
            self._live_dev[key] = (self._live_lo_host.to(device), self._live_hi_host.to(device))
        return self._live_dev[key]

    def conv(self, X: torch.Tensor) -> torch.Tensor:
        lo, hi = self._live(X.device)
        ops.mask_taps_(self.conv1d.weight.data, lo, hi)                      # Q1: re-mask .data every forward
        return ops.conv1d(self.spec, X.contiguous(), self.conv1d.weight, self.conv1d.bias)

    def forward(self, X: torch.Tensor, defer_bn: bool = False) -> torch.Tensor:
        if not defer_bn and inference_mode(self.bn):
            return self.forward_folded(X)
        y = self.conv(X)
        if defer_bn:
            return y
        return batch_norm_act(y, self.bn, self.relu_or_not_at_last_layer)

    def forward_folded(self, X: torch.Tensor, res: Optional[torch.Tensor] = None, relu: Optional[bool] = None) -> torch.Tensor:
        """Inference: conv, BatchNorm (running statistics), optional residual add and ReLU in ONE launch — the
        normalisation is folded into the weights and the bias, the rest is the conv engine's epilogue."""
        lo, hi = self._live(X.device)
        ops.mask_taps_(self.conv1d.weight.data, lo, hi)                      # the reference re-masks in eval mode too
        if not hasattr(self, "_fold"):
            self._fold = _FoldedBN()
        w, b = self._fold.get(self.conv1d, self.bn)
        relu = self.relu_or_not_at_last_layer if relu is None else relu
        return self.spec.forward(X.contiguous(), None, w, None, b, res=res, flags=ops.EPI_RELU if relu else 0)


def inference_mode(*bns: nn.BatchNorm1d) -> bool:
    """Eval-mode BatchNorm with autograd off (the eval pass of utils.py:27-183, the K-way forward of
    multi_source_voting.py:283-291): the normalisation is an affine map per channel that folds into the conv."""
    return (not torch.is_grad_enabled()) and all(not bn.training for bn in bns)


class _FoldedBN:
    """conv → eval-mode BatchNorm as ONE conv:  W' = W·γ/√(σ²+ε),  b' = (b − μ)·γ/√(σ²+ε) + β.

    The fold lives exactly as long as the enclosing ``ops.pack_cache()`` scope — the eval pass (checkpoint.eval_accuracy)
    and the K-way voting forward (voting.collect_logits) open one around all their batches, so an eval loop folds once —
    and is redone on every call outside such a scope.  It is deliberately NOT keyed on tensor version counters: the train
    step updates parameters and running statistics through raw pointers (fst_bn_finalize) and inside replayed hipGraphs,
    neither of which bumps a version, so a version-keyed fold would go stale silently after the first eval."""

    def get(self, conv: nn.Conv1d, bn: nn.BatchNorm1d):
        cache = ops._PACK_CACHE
        key = ("bn_fold", id(self))
        if cache is not None and key in cache:
            return cache[key][0]
        with torch.no_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            wb = ((conv.weight * scale.view(-1, 1, 1)).contiguous(),
                  ((conv.bias - bn.running_mean) * scale + bn.bias).contiguous())
        if cache is not None:
            cache[key] = (wb, self)                  # keeps ``self`` alive: the id cannot be recycled within the scope
        return wb


def batch_norm_act(y: torch.Tensor, bn: nn.BatchNorm1d, relu: bool) -> torch.Tensor:
    if bn.training:
        bn.num_batches_tracked += 1
    return ops.BNActFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, relu, bn.eps,
                             bn.momentum)


class OS_CNN(nn.Module):
    """Classifier: omni-scale layers (all ReLU) → global average pool → Linear (OS_CNN.py:80-110)."""

    def __init__(self, layer_parameter_list: List[LayerSpec], n_class: int, few_shot: bool = False):
        super().__init__()
        self.few_shot = few_shot
        self.layer_parameter_list = layer_parameter_list
        # nothing reads the classifier's masked-tap gradients (GradNorm differentiates the feature extractor's OS_block) and masked
        # weights are re-zeroed every forward: a masked tap's gradient is either not computed (0: the live-tap plans) or the reference's
        # dense value (where the dense many-tap kernel serves the layer: it is the faster of the two)
        self.layer_list = [build_layer_with_layer_parameter(lp, dense_weight_grad=False) for lp in layer_parameter_list]
        for layer in self.layer_list:
            layer.spec.dense_if_fast = True
        self.net = nn.Sequential(*self.layer_list)
        self.averagepool = nn.AdaptiveAvgPool1d(1)
        width = out_channels(layer_parameter_list[-1])
        self.hidden = nn.Linear(width, n_class)
        self.length_before_classification = width

    def forward(self, X: torch.Tensor):
        for layer in self.layer_list:
            X = layer(X)
        X_f = X.mean(dim=-1)
        if not self.few_shot:
            X = ops.linear_act(X_f, self.hidden)
        return X, X_f


class OS_block(nn.Module):
    """Stack of omni-scale layers; the last one's ReLU is optional (OS_CNN.py:117-139)."""

    def __init__(self, layer_parameter_list: List[LayerSpec], relu_or_not_at_last_layer: bool = True):
        super().__init__()
        self.layer_parameter_list = layer_parameter_list
        self.relu_or_not_at_last_layer = relu_or_not_at_last_layer
        n = len(layer_parameter_list)
        self.layer_list = [build_layer_with_layer_parameter(lp, True if i != n - 1 else relu_or_not_at_last_layer)
                           for i, lp in enumerate(layer_parameter_list)]
        self.net = nn.Sequential(*self.layer_list)

    def forward(self, X: torch.Tensor, defer_last_bn: bool = False) -> torch.Tensor:
        last = len(self.layer_list) - 1
        for i, layer in enumerate(self.layer_list):
            X = layer(X, defer_bn=(defer_last_bn and i == last))
        return X


class SampaddingConv1D_BN(nn.Module):
    """"same"-padded Conv1d → BatchNorm1d (the residual shortcut, OS_CNN.py:155-166)."""

    def __init__(self, in_channels: int, out_channels_: int, kernel_size: int):
        super().__init__()
        self.conv1d = nn.Conv1d(in_channels, out_channels_, kernel_size)
        self.bn = nn.BatchNorm1d(out_channels_)
        self.spec = ops.ConvSpec(out_channels_, in_channels, kernel_size, 1, int((kernel_size - 1) / 2))

    def conv(self, X: torch.Tensor) -> torch.Tensor:
        return ops.conv1d(self.spec, X.contiguous(), self.conv1d.weight, self.conv1d.bias)

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        if inference_mode(self.bn):
            return self.forward_folded(X)
        return batch_norm_act(self.conv(X), self.bn, False)

    def forward_folded(self, X: torch.Tensor) -> torch.Tensor:
        if not hasattr(self, "_fold"):
            self._fold = _FoldedBN()
        w, b = self._fold.get(self.conv1d, self.bn)
        return self.spec.forward(X.contiguous(), None, w, None, b)


class Res_OS_layer(nn.Module):
    """relu(BN(shortcut conv) + OS_block(X)) (OS_CNN.py:169-180); the block's last BN, the shortcut's BN, the
    add and the ReLU run as one fused pass."""

    def __init__(self, layer_parameter_list: List[LayerSpec], out_put_channel_numebr: int):
        super().__init__()
        self.layer_parameter_list = layer_parameter_list
        self.net = OS_block(layer_parameter_list, False)
        self.res = SampaddingConv1D_BN(layer_parameter_list[0][0][0], out_put_channel_numebr, 1)

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        if inference_mode(self.net.layer_list[-1].bn, self.res.bn):
            # inference: shortcut conv+BN as one launch, then the block's last conv+BN + shortcut + ReLU as one launch
            layers = self.net.layer_list
            h = X
            for layer in layers[:-1]:
                h = layer(h)                                                 # folded: conv + BN + ReLU in one launch each
            return layers[-1].forward_folded(h, res=self.res.forward_folded(X), relu=True)
        y_block = self.net(X, defer_last_bn=True)
        y_short = self.res.conv(X)
        bn_a, bn_b = self.net.layer_list[-1].bn, self.res.bn
        if bn_a.training:
            bn_a.num_batches_tracked += 1
        if bn_b.training:
            bn_b.num_batches_tracked += 1
        if bn_a.training != bn_b.training or bn_a.eps != bn_b.eps or bn_a.momentum != bn_b.momentum:
            raise RuntimeError("Res_OS_layer: the two BatchNorms must share mode, eps and momentum")
        return ops.BNAddBNReluFn.apply(y_block, bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var,
                                       y_short, bn_b.weight, bn_b.bias, bn_b.running_mean, bn_b.running_var,
                                       bn_a.training, bn_a.eps, bn_a.momentum)


class OS_CNN_res(nn.Module):
    """Residual feature extractor: ``n_layers`` Res_OS_layers, no head (OS_CNN.py:183-220)."""

    def __init__(self, layer_parameter_list: List[LayerSpec], n_layers: int = 1):
        super().__init__()
        self.layer_parameter_list = layer_parameter_list
        self.n_layers = n_layers
        width = out_channels(layer_parameter_list[-1])
        self.net_1 = Res_OS_layer(layer_parameter_list, width)
        self.net_list = [Res_OS_layer(layer_parameter_list_input_change(layer_parameter_list, width), width)
                         for _ in range(n_layers - 1)]
        if n_layers > 1:
            self.net = nn.Sequential(*self.net_list)

    def forward(self, X: torch.Tensor) -> torch.Tensor:
        out = self.net_1(X)
        for layer in self.net_list:
            out = layer(out)
        return out

    def return_last_layer(self) -> nn.Module:
        """The OS_block whose 12 parameters GradNorm differentiates (train_and_test.py:682-690)."""
        return self.net_1.net
