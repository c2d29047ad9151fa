#!/usr/bin/env python3
"""Time the fused WN layer kernels at the metric shape (B=256, L=512, n=120, h=25) for a few dilations.
FST_HIP_LIB picks the library (tools/build_wn_exp.sh variants remove one cost at a time)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops

dev = torch.device("cuda:0")
B, L, n, h = int(os.environ.get("B", 256)), int(os.environ.get("L", 512)), 120, 25
torch.manual_seed(0)
r = lambda *s, k=1.0: torch.randn(*s, device=dev) * k
a, u0 = r(B, n, L), r(B, 2 * h, L)[:, :h]
img = ops.wn_pack_layer(r(2 * n, n, 3, k=.05), r(2 * n, h, 1, k=.1), r(2 * n, k=.1), r(2 * n, k=.1), r(2 * n, n, 1, k=.09), r(2 * n, k=.1), n, h, False)
ts, acts, an, out = torch.empty(B, 2 * n, L, device=dev), torch.empty(B, n, L, device=dev), torch.empty(B, n, L, device=dev), r(B, n, L)
img_b = ops.wn_pack_bwd(r(2 * n, n, k=.09), n, False)
d_a, d_out, dg = r(B, n, L), r(B, n, L), torch.empty(B, 2 * n, L, device=dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out_s = []
for dil in (1, 16, 128):
    out_s.append(f"d{dil}: {timed(lambda: ops.wn_layer_fwd(a, u0, img, ts, acts, an, out, False, False, n, h, dil)):6.1f}")
out_s.append(f"no-acts d16: {timed(lambda: ops.wn_layer_fwd(a, u0, img, ts, None, an, out, False, False, n, h, 16)):6.1f}")
out_s.append(f"bwd: {timed(lambda: ops.wn_layer_bwd(d_a, d_out, ts, img_b, dg, False, n)):6.1f}")
img_d = ops.wn_pack_dgrad(r(2 * n, n, 3, k=.05), r(2 * n, h, 1, k=.1), n, h)
d_u0 = r(B, h, L)
for dil in (1, 16, 128):
    out_s.append(f"dgrad d{dil}: {timed(lambda: ops.wn_layer_dgrad(dg, img_d, d_a, d_u0, n, h, dil)):6.1f}")
print(os.path.basename(os.environ.get("FST_HIP_LIB", "libfst_hip.so")), " us  ", "  ".join(out_s))
