#!/usr/bin/env python3
"""Diagnostics: HIP-event time of fst_gemm against the library GEMM (torch.mm / addmm: hipBLASLt, exact f32) at the dense
products the joint step runs (tools/gemm_census.py), per operand layout."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops

dev = "cuda"


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1000 * e0.elapsed_time(e1) / n


SHAPES = [("dimunif", 12800, 512, 512), ("ad_net", 256, 1024, 1024), ("fd 800-400", 256, 400, 800), ("fd 50-800", 256, 800, 50),
          ("cpc proj", 20992, 192, 50)]
for name, M, N, K in SHAPES:
    x, W, b, g = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    rows = [
        ("fwd  y=xW^T+b,relu", lambda: ops.gemm(x, False, W, False, b, ops.ACT_RELU), lambda: torch.relu(torch.addmm(b, x, W.t()))),
        ("dx = g W          ", lambda: ops.gemm(g, False, W, True), lambda: g @ W),
        ("dW = g^T x        ", lambda: ops.gemm(g, True, x, True), lambda: g.t() @ x),
    ]
    for what, ours, lib in rows:
        a, c = t_us(ours), t_us(lib)
        print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}  {what}  fst_gemm {a:7.1f} us ({2e-6 * M * N * K / a:6.1f} TFLOP/s)   library {c:7.1f} us")
