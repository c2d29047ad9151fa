#!/usr/bin/env python3
"""Diagnostics: which source lines of the package launch the torch (ATen) kernels of ONE eager joint step.

torch.profiler with Python stacks; every op with device time of its own is attributed to the innermost frame inside
feature_level_style_transfer_for_tsc_amd/ (or "autograd/optimizer" when none is on the stack).  Prints launches and device
time per (op, source line), largest first — the list the launch-tail work of DESIGN.md is driven from.
  B=256 L=512 python tools/launch_census.py > gpurun_out/census.txt
"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from bench import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(1234)
B, L = int(os.environ.get("B", 256)), int(os.environ.get("L", 512))
tr = fst.JointTrainer(fst.JointConfig(L_t=L, C_in_t=1, L_s=L, C_in_s=1, n_class_t=4, n_class_s=4), dev)
x_t, y_t = synthetic_batch(B, 1, L, 4, dev, 1000)
x_s, y_s = synthetic_batch(B, 1, L, 4, dev, 2000)
for _ in range(2):
    tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
    torch.cuda.synchronize()

agg = defaultdict(lambda: [0, 0.0])
n_kernels = 0
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks:
        continue
    n_kernels += len(ks)
    where = None
    for fr in (e.stack or []):
        if "feature_level_style_transfer_for_tsc_amd" in fr:
            where = fr.split("feature_level_style_transfer_for_tsc_amd/")[-1]
            break
    if where is None:                        # no Python stack on this event: the chain of enclosing ops / autograd nodes instead
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name)
            q = q.cpu_parent
        where = " < ".join(chain) if chain else "(top level)"
    key = (e.name, where)
    agg[key][0] += len(ks)
    agg[key][1] += sum(k.duration for k in ks)
print(f"kernels launched by torch ops in one eager step: {n_kernels}")
tot_n = tot_us = 0
for (name, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    tot_n += n
    tot_us += us
    print(f"{n:5d} launches {us:9.1f} us  {name:36s} {where}")
print(f"total {tot_n} launches, {tot_us / 1e3:.2f} ms of device time")
