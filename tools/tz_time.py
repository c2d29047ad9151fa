#!/usr/bin/env python3
"""Diagnostics: HIP-event time of the dense many-tap weight gradient (fst_dense_tap_wgrad) at the bench's three shapes.
Cost removal: build a diagnostic library with tools/build_tz_exp.sh <mask> (1 no MFMAs, 2 no split pass, 4 no LDS-DMA, 8 no fragment
reads, 16 no slab stores, 32 no stages) and run with FST_HIP_LIB=build/exp/libfst_tzexp<mask>.so (the first column names the library)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops

dev = "cuda"
B, L = 256, 512
out = [os.path.basename(os.environ.get("FST_HIP_LIB", "libfst_hip.so"))]
for (M, C, K) in ((225, 25, 89), (25, 50, 89), (25, 1, 89)):
    x, dy, dw = torch.randn(B, C, L, device=dev), torch.randn(B, M, L, device=dev), torch.empty(M, C, K, device=dev)
    fn = lambda: ops.dense_tap_wgrad(dy, x, dw, M, C, K, (K - 1) // 2)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{M}x{C}x{K}: {100 * e0.elapsed_time(e1):7.1f} us")
print("  ".join(out))
