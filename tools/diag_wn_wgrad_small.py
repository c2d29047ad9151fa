#!/usr/bin/env python3
"""Diagnostics: weight gradients of a small WN stack (n=16, h=5, 3 layers, B=512, L=256) behind the one-launch forward and behind
the per-layer launches, twice each, per parameter segment against fp64."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from feature_level_style_transfer_for_tsc_amd import ops
from test_gpu_full_size import _wn_reference_f64, _rnd

for (n, h, Bq, L, nl, stack_bwd) in ((16, 5, 512, 256, 3, "1"), (16, 5, 512, 256, 3, "0"), (16, 5, 256, 256, 3, "1"), (16, 5, 512, 64, 3, "1"),
                                     (16, 5, 64, 256, 3, "1"), (16, 5, 5, 64, 3, "1"), (48, 5, 512, 256, 3, "1"), (16, 5, 512, 512, 3, "1")):
    os.environ["FST_WN_STACK"] = stack_bwd
    os.environ["FST_WN_STACK_FWD"] = "0"
    g = torch.Generator(device="cuda").manual_seed(n * 17 + L + nl)
    S = ops.WNSpecs(h, n, nl)
    ws = []
    for j, sh in enumerate(S.shapes):
        fan = sh[1] * sh[2] if len(sh) == 3 else 1
        ws.append(_rnd(g, *sh, k=(1.0 / fan ** 0.5 if len(sh) == 3 else 0.1)))
    flat = S.flatten(ws)
    x = _rnd(g, Bq, 2 * h, L)
    do = _rnd(g, Bq, 2 * h, L)
    _, _, dw_ref = _wn_reference_f64(S, x[:, :h], flat, do)
    for rep in range(2):
        u0 = x[:, :h].detach().requires_grad_(True)
        fl = flat.detach().clone().requires_grad_(True)
        timer = ops.KernelTimer(); ops.KERNEL_TIMER = timer
        with ops.pack_cache():
            o = ops.WNFn.apply(S, u0, fl)
            d_u0, d_fl = torch.autograd.grad(o, (u0, fl), do)
        ops.KERNEL_TIMER = None
        errs = []
        for i, (lo, hi) in enumerate(zip(S.offsets[:-1], S.offsets[1:])):
            e = float((d_fl[lo:hi].double() - dw_ref[lo:hi]).abs().max() / max(1e-30, float(dw_ref[lo:hi].abs().max())))
            if e > 1e-3: errs.append(f"{i}:{e:.1e}")
        print("shape", (n, h, Bq, L, nl), "FST_WN_STACK", stack_bwd, "bad segments:", " ".join(errs) or "none", "| kernels:", ",".join(sorted(k.split("(")[0][:28] for k in timer.summary() if "wgrad" in k or "nt_" in k)))
