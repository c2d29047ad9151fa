#!/usr/bin/env python3
"""Diagnostics: the small aten ops of one eager joint step (metric config), counted per input shape."""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key in ("aten::div", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::copy_", "aten::sum", "aten::cat", "aten::mul"):
        rows.append((ev.count, ev.key, str(ev.input_shapes)[:150]))
rows.sort(key=lambda r: -r[0])
for c, k, sh in rows[:60]:
    print(f"{c:5d} {k:12s} {sh}")
