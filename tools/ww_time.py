#!/usr/bin/env python3
"""Diagnostics: HIP-event time of the two time-as-k weight-gradient launches (csrc/wn_wgrad.hip) at the bench's shapes, main +
reduce kernel together.  FST_WW_EXP=<mask> (read by the library at its first launch) removes one cost at a time (timing only):
1 LDS-DMA pieces from the zero block, 2 no k-step arithmetic, 4 no LDS-DMA at all.
    for e in 0 1 2 3 4 6; do FST_WW_EXP=$e python tools/ww_time.py; done"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops

dev = "cuda"
B, L, n, h = 256, int(os.environ.get("L", 512)), 120, 25
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
a = ops.empty_with_slack(B, n, L, dev); a.copy_(rnd(B, n, L))
dg, u0, ts, d_a, d_out = rnd(B, 2 * n, L), rnd(B, 2 * h, L)[:, :h], rnd(B, 2 * n, L), rnd(B, n, L), rnd(B, n, L)
dw_in, dw_cond, dw_rs = torch.empty(2 * n, n, 3, device=dev), torch.empty(2 * n, h, 1, device=dev), torch.empty(2 * n, n, 1, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


NS = int(os.environ.get("SETS", 1))              # operand sets summed by one launch (the applications of a WN)
if NS > 1:
    def more(t):
        out = [t]
        for _ in range(NS - 1):
            c = ops.empty_with_slack(*t.shape, dev) if t is a else torch.empty_like(t)
            out.append(c.copy_(t))
        return out
    a, dg, ts, d_a, d_out = more(a), more(dg), more(ts), more(d_a), more(d_out)
    u0 = [u0] + [rnd(B, 2 * h, L)[:, :h] for _ in range(NS - 1)]
out = [f"FST_WW_EXP={os.environ.get('FST_WW_EXP', '0')} sets={NS}"]
for dil in (1, 4, 128):
    out.append(f"in dil={dil}: {timed(lambda: ops.wn_wgrad_in(dg, a, u0, dw_in, dw_cond, n, h, dil)):6.1f} us")
out.append(f"rs: {timed(lambda: ops.wn_wgrad_rs(d_a, d_out, ts, dw_rs, False, n)):6.1f} us")
print("  ".join(out))
