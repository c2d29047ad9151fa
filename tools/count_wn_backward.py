#!/usr/bin/env python3
"""Diagnostics: how many times WNFn.backward runs in each backward pass of one joint step (small golden config)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import ops

g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "joint_small.npz"), allow_pickle=False))
meta = json.loads(str(g["meta"]))
tup = lambda lp: [[tuple(t) for t in l] for l in lp]
sub = lambda prefix: {k[len(prefix):]: torch.tensor(v) for k, v in g.items() if k.startswith(prefix)}
cfg = fst.JointConfig(L_t=meta["L_t"], C_in_t=meta["C_in_t"], L_s=meta["L_s"], C_in_s=meta["C_in_s"], n_class_t=meta["ncls_t"],
                      n_class_s=meta["ncls_s"], nf_channels=meta["nf"][2], cpc_hidden=meta["cpc"][1], cdan_dim=64, ad_hidden=32, dropout_p=0.0)
tr = fst.JointTrainer(cfg, "cuda", fe_t_spec=tup(meta["lp_t"]), clf_spec=tup(meta["lp_clf"]), fe_s_spec=tup(meta["lp_s"]))
tr.load_params({name: sub(f"sd0.{name}.") for name in tr.MODULES}, [torch.tensor(g["m0"]), torch.tensor(g["m1"])])
args = [torch.tensor(g[f"s0.{k}"], device="cuda") for k in ("x_t", "y_t", "x_s", "y_s")]

calls = []
orig_bwd = ops.WNFn.backward
def counting(ctx, do):
    calls.append(("partial" if ops._PARTIAL_BACKWARD else "main", tuple(ctx.needs_input_grad[:3])))
    return orig_bwd(ctx, do)
ops.WNFn.backward = staticmethod(counting)
orig_grad = torch.autograd.grad
marks = []
def marked(outputs, inputs, *a, **k):
    marks.append(len(calls))
    return orig_grad(outputs, inputs, *a, **k)
torch.autograd.grad = marked
tr.step(*args, epoch=0, t_samples=(2, 5))
torch.cuda.synchronize()
n_flows = 3
print("WNFn.backward calls:", len(calls), "=", len(calls) / n_flows, "WaveGlow traversals")
print("main:", sum(1 for c in calls if c[0] == "main") / n_flows, " partial:", sum(1 for c in calls if c[0] == "partial") / n_flows)
b = marks + [len(calls)]
print("per autograd.grad call:", [(b[i + 1] - b[i]) / n_flows for i in range(len(marks))])
