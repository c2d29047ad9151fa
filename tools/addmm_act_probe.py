#!/usr/bin/env python3
"""Diagnostics: does torch._addmm_activation (bias + ReLU in the hipBLASLt epilogue) run as ONE kernel on this ROCm build?"""
import torch
from torch.profiler import ProfilerActivity, profile
dev = "cuda"
x = torch.randn(12800, 512, device=dev); W = torch.randn(512, 512, device=dev); b = torch.randn(512, device=dev)
for name, fn in (("addmm+relu", lambda: torch.relu(torch.addmm(b, x, W.t()))),
                 ("_addmm_activation", lambda: torch._addmm_activation(b, x, W.t(), use_gelu=False))):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        y = fn(); torch.cuda.synchronize()
    ks = [(e.name[:70], e.device_time) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(name, ks)
ref = torch.relu(x.double() @ W.double().t() + b.double())
print("max err", float((torch._addmm_activation(b, x, W.t(), use_gelu=False).double() - ref).abs().max() / ref.abs().max()))
