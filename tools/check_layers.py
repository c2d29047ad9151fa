#!/usr/bin/env python3
"""Diagnostics: per-layer conv forward / data-gradient / weight-gradient errors at a given (B, L) vs fp64 torch."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops
from feature_level_style_transfer_for_tsc_amd.structure import generate_layer_parameter_list, out_channels, row_live_ranges

B, L, CIN = int(os.environ.get("B", 2)), int(os.environ.get("L", 5000)), int(os.environ.get("CIN", 9))
DEV = "cuda"
lp = generate_layer_parameter_list(1, min(L // 4, 89), [1024 * CIN, 229376], CIN)
g = torch.Generator().manual_seed(0)


def ref_conv(x, w, dil, pl, nt):
    halo = (nt - 1) * dil
    return F.conv1d(F.pad(x.double(), (pl, halo - pl)), w.double(), dilation=dil)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(1e-9, float(b.abs().max())))


for name, layer in (("L0", lp[0]), ("L1", lp[1]), ("L2", lp[2]), ("short", [(CIN, 50, 1)])):
    C0, kmax, M = layer[0][0], layer[-1][2], out_channels(layer)
    live = row_live_ranges(layer) if name != "short" else None
    w = torch.randn(M, C0, kmax, generator=g, dtype=torch.float64) / (C0 * 3) ** 0.5
    if live:
        mask = torch.zeros_like(w)
        for m, (lo, hi) in enumerate(live):
            mask[m, :, lo:hi] = 1
        w = w * mask
    w.requires_grad_(True)
    x = torch.randn(B, C0, L, generator=g, dtype=torch.float64, requires_grad=True)
    pl = int((kmax - 1) / 2)
    y = ref_conv(x, w, 1, pl, kmax)
    dy = torch.randn(B, M, L, generator=g, dtype=torch.float64)
    (y * dy).sum().backward()
    spec = ops.ConvSpec(M, C0, kmax, 1, pl, row_live=live)
    f = lambda t: t.detach().float().to(DEV)
    yd = spec.forward(f(x), None, f(w), None, None)
    dx = spec.grad_x0(f(dy), f(w))
    dw, _ = spec.grad_w(f(x), None, f(dy))
    if live:
        xd_, dyd_ = x.detach(), dy
        halo = kmax - 1
        dwd_ref = torch.autograd.grad(ref_conv(xd_, wd := torch.zeros(M, C0, kmax, dtype=torch.float64, requires_grad=True), 1, pl, kmax), wd, dyd_)[0]
        sd = ops.ConvSpec(M, C0, kmax, 1, pl, row_live=live, dense_dw=True)
        dwd, _ = sd.grad_w(f(x), None, f(dy))
        print(f"   dense dW {rel(dwd, dwd_ref):.2e}")
    e = (dx.double().cpu() - x.grad).abs()
    bad_t = torch.nonzero(e.amax(dim=(0, 1)) > 1e-3 * x.grad.abs().max()).flatten()
    print(f"{name}: M={M} C0={C0} K={kmax}  fwd {rel(yd, y):.2e}  dx {rel(dx, x.grad):.2e}  dw {rel(dw, w.grad):.2e}   bad t: {bad_t[:8].tolist()}..{bad_t[-4:].tolist() if len(bad_t) else ''} n={len(bad_t)}")
