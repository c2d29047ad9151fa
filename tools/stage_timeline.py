#!/usr/bin/env python3
"""Diagnostics: where a wave of conv_gemm_bf3_kernel spends its cycles (needs the FST_STAMPS build:
tools/build_stamps.sh, then FST_HIP_LIB=build/exp/libfst_hip_stamps.so python tools/stage_timeline.py)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import _lib, ops

lib = _lib.load()
lib.fst_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
B, L, n, h = 256, 512, 120, 25
names = ["prologue", "next+vmcnt", "barrier", "dma issue", "B frags", "A+mfma", "epilogue", "whole wave"]


def report(title, waves, stages):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.fst_debug_stamps(buf, 1)
    tot = buf[7] / waves
    print(f"== {title}: {tot:9.0f} cycles/wave, {stages} stages")
    for k in range(7):
        per = buf[k] / waves
        print(f"   {names[k]:12s} {per:9.0f} cyc/wave  {100 * per / tot:5.1f}%   {per / (stages if 1 <= k <= 5 else 1):8.0f} per {'stage' if 1 <= k <= 5 else 'wave'}")


torch.manual_seed(0)
spec = ops.ConvSpec(2 * n, n, 3, 8, 8, C1=h)                                  # WN in_layer
a, u0 = torch.randn(B, n, L, device=dev), torch.randn(B, h, L, device=dev)
w0, w1, bias = torch.randn(2 * n, n, 3, device=dev) * .05, torch.randn(2 * n, h, 1, device=dev) * .05, torch.zeros(2 * n, device=dev)
for it in range(2):
    lib.fst_debug_stamps(None, 1)
    spec.forward(a, u0, w0, w1, bias)
report("in_layer forward  pipe<8,1>", 1024 * 4, 26)
rs = ops.ConvSpec(2 * n, n)
acts, out = torch.randn(B, n, L, device=dev), torch.zeros(B, n, L, device=dev)
wr = torch.randn(2 * n, n, 1, device=dev) * .05
for it in range(2):
    lib.fst_debug_stamps(None, 1)
    rs.forward(acts, None, wr, None, bias, y=torch.empty_like(a), res=a, y2=out, msplit=n, flags=ops.EPI_ACC2)
report("res_skip forward  pipe<8,1>", 1024 * 4, 8)
dg = torch.randn(B, 2 * n, L, device=dev)
d_u0 = torch.zeros(B, h, L, device=dev)
for it in range(2):
    lib.fst_debug_stamps(None, 1)
    spec.grad_x01(dg, w0, w1, a, d_u0)
report("in_layer dx01     pipe<4,2> (two M-groups mixed)", 1024 * 4, 30)
