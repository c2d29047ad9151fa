#!/usr/bin/env python3
"""Diagnostics: which Python call sites issue the most aten ops in one eager joint step (metric config)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from bench import synthetic_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
B, L = int(os.environ.get("B", 64)), 512
tr = fst.JointTrainer(fst.JointConfig(L_t=L, C_in_t=1, L_s=L, C_in_s=1, n_class_t=4, n_class_s=4), dev)
x_t, y_t = synthetic_batch(B, 1, L, 4, dev, 1000); x_s, y_s = synthetic_batch(B, 1, L, 4, dev, 2000)
for _ in range(2):
    tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
tab = prof.key_averages(group_by_stack_n=8)
rows = []
for ev in tab:
    if ev.key in ("aten::div", "aten::div_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::copy_"):
        rows.append((ev.count, ev.key, [s for s in ev.stack if "site-packages/torch/autograd" not in s][:6]))
rows.sort(key=lambda r: -r[0])
for c, k, st in rows[:28]:
    print(f"{c:5d} {k:12s} " + " <- ".join(x.split("/")[-1][:60] for x in st[:4]))
