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

# RandomLayer (C_DAN.py:21): x [256, 25600] · R0 [25600, 1024] and its data gradient dy [256, 1024] · R0ᵀ
x, R0, dy = torch.randn(256, 25600, device=dev), torch.randn(25600, 1024, device=dev), torch.randn(256, 1024, device=dev)
R0t = R0.t().contiguous()
for what, ours, lib in [("x R0   (R0 k-major)     ", lambda: ops.gemm(x, False, R0, True), lambda: x @ R0),
                        ("x R0   (R0^T contiguous)", lambda: ops.gemm(x, False, R0t, False), lambda: x @ R0),
                        ("dy R0^T (R0 rows = n)   ", lambda: ops.gemm(dy, False, R0, False), lambda: dy @ R0.t()),
                        ("nt_gemm fwd             ", lambda: ops.nt_gemm(x, R0t), lambda: x @ R0),
                        ("nt_gemm dgrad           ", lambda: ops.nt_gemm(dy, R0), lambda: dy @ R0.t())]:
    a, c = t_us(ours), t_us(lib)
    print(f"random layer  {what}  ours {a:7.1f} us   library {c:7.1f} us")
