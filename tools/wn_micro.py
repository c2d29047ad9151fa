#!/usr/bin/env python3
"""Micro-benchmark of one WaveGlow WN (h=25, n=120, 8 layers) at the metric batch (B=256, L=512): forward +
backward, with per-shape HIP-event timings of every conv-engine launch.  Diagnostics only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, L = int(os.environ.get("B", 256)), int(os.environ.get("L", 512))
wn = fst.WN(25, 8, 120, 3).to(dev)
wn.end.weight.data.normal_(0, 0.05)
u = torch.randn(B, 50, L, device=dev, requires_grad=True)
reps = int(os.environ.get("REPS", 3))
for it in range(reps + 1):
    if it == 1:
        timer = ops.KernelTimer(detail=True)
        ops.KERNEL_TIMER = timer
        torch.cuda.synchronize(); t0 = time.perf_counter()
    o = wn(u[:, :25])
    o.square().sum().backward()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
ops.KERNEL_TIMER = None
print(f"WN fwd+bwd: {dt*1e3:.2f} ms wall")
tot = 0.0
for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    tot += v["total_ms"]
    print(f"{k:75s} n={v['launches']//reps:3d}/it avg={v['avg_us']:8.1f} us  {v['flops']/(v['total_ms']*1e-3)/1e12:6.1f} TF  tot={v['total_ms']/reps:7.2f} ms/it")
print(f"conv engine total {tot/reps:.2f} ms/it")
