#!/usr/bin/env python3
"""The three kernels BASELINE.json's north_star names besides the step rate, alone, at the metric batch (B=256, L=512):
the omni-scale Conv1d sweep (OS_CNN_res forward, train-mode BatchNorm), the CPC cross-Gram (CPCNceFn forward) and the
CDAN random-layer GEMM (ops.nt_gemm with RandomLayer's epilogue, and its data gradient).  Run under rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES
passes) for the counter evidence in profiles/r02_*; prints HIP-event timings otherwise."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L, C = int(os.environ.get("B", 256)), int(os.environ.get("L", 512)), 50
reps = int(os.environ.get("REPS", 5))
fe_spec, _ = fst.specs_for(L, 1)
fe = fst.OS_CNN_res(fe_spec).to(dev).train()
x = torch.randn(B, 1, L, device=dev)
feat = torch.randn(B, C, L, device=dev)
T = L // 2
pred = torch.randn(T, B, C, device=dev) * 0.3
xf = torch.randn(B, C * L, device=dev)
R0 = torch.randn(C * L, 1024, device=dev)


def timed(name, fn, flops, nbytes):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:28s} {ms * 1e3:8.1f} us  {flops / ms / 1e9:8.1f} TFLOP/s  {nbytes / ms / 1e6:8.1f} GB/s (algorithmic)")


with torch.no_grad():
    macs = 964 + 216900 + 16875 + 50
    timed("omni_scale_fe_forward", lambda: fe(x), 2.0 * macs * B * L, 4.0 * (1 + C) * B * L)
    timed("cpc_cross_gram", lambda: ops.CPCNceFn.apply(feat, pred, 7, T), 2.0 * T * B * B * C, 4.0 * (B * C * T + T * B * C))
    R0t = R0.t().contiguous()
    R1 = torch.randn(4, 1024, device=dev)
    pr = torch.softmax(torch.randn(B, 4, device=dev), 1)
    dyr = torch.randn(B, 1024, device=dev)
    timed("cdan_random_layer_gemm", lambda: ops.nt_gemm(xf, R0t, (pr, R1, 1.0 / 32.0)), 2.0 * B * C * L * 1024,
          4.0 * (R0.numel() + xf.numel() + B * 1024))
    timed("cdan_random_layer_dgrad", lambda: ops.nt_gemm(dyr, R0), 2.0 * B * C * L * 1024, 4.0 * (R0.numel() + xf.numel() + B * 1024))
