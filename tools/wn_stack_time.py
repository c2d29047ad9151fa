#!/usr/bin/env python3
"""Diagnostics: the backward of one WN stack (n=120, h=25, 8 layers) at the bench shape, as one persistent launch
(fst_wn_stack_bwd) against one launch pair per layer; without weight gradients (GradNorm's partial passes) and with the
operands of the weight gradients kept (the full pass; the weight-gradient kernels themselves are not timed here)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, L, n, h, nl = int(os.environ.get("WS_B", 256)), int(os.environ.get("WS_L", 512)), 120, 25, 8
torch.manual_seed(0)
S = ops.WNSpecs(h, n, nl)
ws = []
for sh in S.shapes:
    fan = sh[1] * sh[2] if len(sh) == 3 else 1
    ws.append(torch.randn(*sh, device=dev) * (1.0 / fan ** 0.5 if len(sh) == 3 else 0.1))
flat = S.flatten(ws).requires_grad_(True)
x = torch.randn(B, 2 * h, L, device=dev)
do = torch.randn(B, 2 * h, L, device=dev)
reps = int(os.environ.get("WS_REPS", 10))
for mode in ("0", "1"):
    os.environ["FST_WN_STACK"] = mode
    with ops.pack_cache():
        u0 = x[:, :h].detach().requires_grad_(True)
        o = ops.WNFn.apply(S, u0, flat)
        for partial in (True, False):
            def run():
                if partial:
                    with ops.partial_backward():
                        torch.autograd.grad(o, (u0,), do, retain_graph=True)
                else:
                    # the data path of the full pass: weight-gradient operands kept, their kernels deferred to a pool nobody reads
                    sv = o.grad_fn.saved_tensors
                    d_u0 = torch.zeros(B, h, L, device=dev)
                    ops._wn_backward(S, True, sv, do, d_u0, True, ops.WNGradPool())
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            print(f"FST_WN_STACK={mode} {'partial (no weight gradients)' if partial else 'full data path (operands kept)  '}: "
                  f"{e0.elapsed_time(e1) * 1e3 / reps:9.1f} us per stack backward = {e0.elapsed_time(e1) * 1e3 / reps / nl:7.1f} us per layer", flush=True)
