#!/usr/bin/env python3
"""Diagnostics: which backward passes of one joint step compute the dense (Q1) weight gradient of the shared omni-scale layers."""
import os, sys, traceback, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import ops
from bench import synthetic_batch

dev = torch.device("cuda:0")
B, L = int(os.environ.get("B", 32)), 512
tr = fst.JointTrainer(fst.JointConfig(L_t=L, C_in_t=1, L_s=L, C_in_s=1, n_class_t=4, n_class_s=4), dev)
x_t, y_t = synthetic_batch(B, 1, L, 4, dev, 1000)
x_s, y_s = synthetic_batch(B, 1, L, 4, dev, 2000)
tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
calls = collections.Counter()
orig = ops.ConvSpec.grad_w


def grad_w(self, *a, **k):
    if getattr(self, "dense_dw", False):
        st = [f.name for f in traceback.extract_stack() if "step.py" in f.filename]
        calls[(self.M, self.C0, ops._PARTIAL_BACKWARD, tuple(st[-2:]))] += 1
    return orig(self, *a, **k)


ops.ConvSpec.grad_w = grad_w
tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(10, 20))
for k, v in sorted(calls.items(), key=str):
    print(v, k)
