#!/usr/bin/env python3
"""Diagnostics: where a wave of wn_stack_bwd_kernel spends its cycles (needs the FST_STAMPS build:
tools/build_stamps.sh, then FST_HIP_LIB=build/exp/libfst_hip_stamps.so python tools/wn_stack_timeline.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.fst_debug_wn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
B, L, n, h, nl = 256, 512, 120, 25, 8
names = ["A: first loads + priming", "A: GEMM loop", "A: gate epilogue", "A->B drain + barrier", "B: acc loads + priming",
         "B: GEMM loop", "B: epilogue stores", "B->A drain + barrier"]
torch.manual_seed(0)
S = ops.WNSpecs(h, n, nl)
ws = []
for sh in S.shapes:
    fan = sh[1] * sh[2] if len(sh) == 3 else 1
    ws.append(torch.randn(*sh, device=dev) * (1.0 / fan ** 0.5 if len(sh) == 3 else 0.1))
flat = S.flatten(ws).requires_grad_(True)
x, do = torch.randn(B, 2 * h, L, device=dev), torch.randn(B, 2 * h, L, device=dev)
with ops.pack_cache():
    u0 = x[:, :h].detach().requires_grad_(True)
    o = ops.WNFn.apply(S, u0, flat)
    for partial in (True, False):
        for it in range(3):
            lib.fst_debug_wn_stamps(None, 1)
            if partial:
                with ops.partial_backward():
                    torch.autograd.grad(o, (u0,), do, retain_graph=True)
            else:
                ops._wn_backward(S, True, o.grad_fn.saved_tensors, do, torch.zeros(B, h, L, device=dev), True, ops.WNGradPool())
            torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 12)()
        lib.fst_debug_wn_stamps(buf, 1)
        waves = B * 8
        tot = buf[9] / waves
        print(f"== {'partial pass' if partial else 'full pass (row sums, operands kept)'}: {tot:.0f} cycles per wave, {tot / nl:.0f} per layer")
        for k in range(8):
            per = buf[k] / waves
            print(f"   {names[k]:26s} {per:9.0f} cyc/wave {100 * per / tot:5.1f}%  {per / nl:8.0f} per layer")
