#!/usr/bin/env python3
"""Diagnostics: per-shape conv-engine time of ONE eager joint step at the metric size (HIP events per launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import ops
from bench import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(1234)
B, L = int(os.environ.get("B", 256)), int(os.environ.get("L", 512))
tr = fst.JointTrainer(fst.JointConfig(L_t=L, C_in_t=1, L_s=L, C_in_s=1, n_class_t=4, n_class_s=4), dev)
x_t, y_t = synthetic_batch(B, 1, L, 4, dev, 1000)
x_s, y_s = synthetic_batch(B, 1, L, 4, dev, 2000)
for _ in range(2):
    tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
timer = ops.KernelTimer(detail=True)
ops.KERNEL_TIMER = timer
tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
torch.cuda.synchronize()
ops.KERNEL_TIMER = None
tot = 0.0
for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
    tot += v["total_ms"]
    if v["total_ms"] > 0.4:
        print(f"{k:78s} n={v['launches']:4d} avg={v['avg_us']:8.1f} us {v['flops']/(v['total_ms']*1e-3)/1e12:6.1f} TF "
              f"{v['bytes']/(v['total_ms']*1e-3)/1e12:5.2f} TB/s tot={v['total_ms']:6.2f} ms")
print(f"conv engine total {tot:.1f} ms")
