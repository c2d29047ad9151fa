#!/usr/bin/env python3
"""Diagnostics: which source lines of the package ask for zero-filled / filled tensors or device copies during one eager joint step."""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from bench import synthetic_batch

dev = torch.device("cuda:0")
B, L = int(os.environ.get("B", 32)), 512
tr = fst.JointTrainer(fst.JointConfig(L_t=L, C_in_t=1, L_s=L, C_in_s=1, n_class_t=4, n_class_s=4), dev)
x_t, y_t = synthetic_batch(B, 1, L, 4, dev, 1000)
x_s, y_s = synthetic_batch(B, 1, L, 4, dev, 2000)
for _ in range(2):
    tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
calls = collections.Counter()


def where():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "feature_level_style_transfer_for_tsc_amd" in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno} {f.name}"
    return "(outside the package)"


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        calls[(name, where())] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for n in ("zeros", "zeros_like", "ones", "ones_like", "full", "full_like"):
    wrap(torch, n)
for n in ("zero_", "fill_", "copy_", "clone", "contiguous"):
    wrap(torch.Tensor, n)
tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
for (name, w), v in sorted(calls.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{v:5d} {name:12s} {w}")
