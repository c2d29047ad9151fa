#!/usr/bin/env python3
"""Diagnostics: the omni-scale layer-1 convolution (225 rows x 25 channels x 89 taps, live taps only) alone at B=256, L=512:
forward on the window kernel (conv_win_bf3_kernel), us / useful TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst  # noqa: E402
from feature_level_style_transfer_for_tsc_amd import ops  # noqa: E402
from feature_level_style_transfer_for_tsc_amd.structure import out_channels, row_live_ranges  # noqa: E402

dev = torch.device("cuda:0")
B, L = int(os.environ.get("B", 256)), int(os.environ.get("L", 512))
fe_spec, _ = fst.specs_for(L, 1)
layer = fe_spec[1]
C0, kmax, M = layer[0][0], layer[-1][2], out_channels(layer)
live = row_live_ranges(layer)
macs = sum(hi - lo for lo, hi in live) * C0
torch.manual_seed(0)
w = torch.randn(M, C0, kmax, device=dev) / (C0 * 3) ** 0.5
for m, (lo, hi) in enumerate(live):
    w[m, :, :lo] = 0
    w[m, :, hi:] = 0
x = torch.randn(B, C0, L, device=dev)
spec = ops.ConvSpec(M, C0, kmax, 1, int((kmax - 1) / 2), row_live=live)
with ops.pack_cache():
    for _ in range(3):
        y = spec.forward(x, None, w, None, None)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = spec.forward(x, None, w, None, None)
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x[:2].double(), (int((kmax - 1) / 2), kmax - 1 - int((kmax - 1) / 2))), w.double())
err = float((y[:2].double() - ref).abs().max() / ref.abs().max())
print(f"FST_WIN_NB={os.environ.get('FST_WIN_NB', '-')} FST_WIN_RESIDENT_KB={os.environ.get('FST_WIN_RESIDENT_KB', '-')}: "
      f"{us:8.1f} us  {2.0 * macs * B * L / us / 1e6:7.1f} useful TFLOP/s  rel err {err:.1e}")
