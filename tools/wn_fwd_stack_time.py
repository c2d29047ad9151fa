#!/usr/bin/env python3
"""Diagnostics: HIP-event time of the forward of one WN stack (8 layers, n=120, h=25, B=256, L=512) as per-layer launches
(FST_WN_STACK_FWD=0) and as the one persistent launch, with the start stagger of every other workgroup swept
(FST_WN_FWD_STAGGER units of 3.4 us; read once per process: run once per value)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops

B, L, n, h, nl = int(os.environ.get("B", 256)), int(os.environ.get("L", 512)), 120, 25, 8
S = ops.WNSpecs(h, n, nl)
g = torch.Generator(device="cuda").manual_seed(0)
ws = [torch.randn(*sh, device="cuda", generator=g) * (1.0 / (sh[1] * sh[2]) ** 0.5 if len(sh) == 3 else 0.1) for sh in S.shapes]
flat = S.flatten(ws)
u0 = torch.randn(B, h, L, device="cuda", generator=g)
for mode in ("0", "1"):
    os.environ["FST_WN_STACK_FWD"] = mode
    with ops.pack_cache(), torch.no_grad():
        for _ in range(3):
            ops._wn_forward(S, u0, flat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            ops._wn_forward(S, u0, flat)
        e1.record(); torch.cuda.synchronize()
    print(f"FST_WN_STACK_FWD={mode} stagger={os.environ.get('FST_WN_FWD_STAGGER', 'default')}: {100 * e0.elapsed_time(e1):8.1f} us per stack forward "
          f"(start conv + {nl} layers + end conv) = {100 * e0.elapsed_time(e1) / nl:6.1f} us per layer")
