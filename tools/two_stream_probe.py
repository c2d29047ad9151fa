#!/usr/bin/env python3
"""Diagnostics: do two independent chains of fused WN layer launches (the forward passes of the target and the source batch)
run faster side by side on two streams than one after the other?  Every workgroup of a launch goes through the same phases at
the same time (main loop, then an HBM burst); a second launch in flight fills one's burst with the other's MFMA time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import ops

dev = torch.device("cuda:0")
B, L, n, h = 256, 512, 120, 25
torch.manual_seed(0)
r = lambda *s, k=1.0: torch.randn(*s, device=dev) * k


def make():
    d = {}
    d["a"], d["u0"] = r(B, n, L), r(B, 2 * h, L)[:, :h]
    d["img"] = ops.wn_pack_layer(r(2 * n, n, 3, k=.05), r(2 * n, h, 1, k=.1), r(2 * n, k=.1), r(2 * n, k=.1), r(2 * n, n, 1, k=.09), r(2 * n, k=.1), n, h, False)
    d["ts"], d["an"], d["out"] = torch.empty(B, 2 * n, L, device=dev), torch.empty(B, n, L, device=dev), r(B, n, L)
    d["img_b"] = ops.wn_pack_bwd(r(2 * n, n, k=.09), n, False)
    d["d_a"], d["d_out"], d["dg"] = r(B, n, L), r(B, n, L), torch.empty(B, 2 * n, L, device=dev)
    d["img_d"] = ops.wn_pack_dgrad(r(2 * n, n, 3, k=.05), r(2 * n, h, 1, k=.1), n, h)
    d["d_u0"] = r(B, h, L)
    return d


X, Y = make(), make()
kinds = {
    "fwd": lambda d: ops.wn_layer_fwd(d["a"], d["u0"], d["img"], d["ts"], None, d["an"], d["out"], False, False, n, h, 16),
    "bwd": lambda d: ops.wn_layer_bwd(d["d_a"], d["d_out"], d["ts"], d["img_b"], d["dg"], False, n),
    "dgrad": lambda d: ops.wn_layer_dgrad(d["dg"], d["img_d"], d["d_a"], d["d_u0"], n, h, 16),
    "bwd+dgrad": lambda d: (ops.wn_layer_bwd(d["d_a"], d["d_out"], d["ts"], d["img_b"], d["dg"], False, n),
                            ops.wn_layer_dgrad(d["dg"], d["img_d"], d["d_a"], d["d_u0"], n, h, 16)),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
reps = 16
for name, fn in kinds.items():
    for _ in range(2):
        fn(X); fn(Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(X); fn(Y)
    e1.record(); torch.cuda.synchronize()
    serial = e0.elapsed_time(e1) * 1e3 / (2 * reps)
    # captured as one graph with two parallel branches (as a captured train step would hold them)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        for _ in range(reps):
            fn(X)
        with torch.cuda.stream(s2):
            for _ in range(reps):
                fn(Y)
        cur.wait_stream(s2)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    par = e0.elapsed_time(e1) * 1e3 / (2 * reps)
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gs):
        for _ in range(reps):
            fn(X); fn(Y)
    gs.replay(); torch.cuda.synchronize()
    e0.record(); gs.replay(); e1.record(); torch.cuda.synchronize()
    ser_g = e0.elapsed_time(e1) * 1e3 / (2 * reps)
    print(f"{name:10s} eager serial {serial:6.1f} us/launch-set   graph serial {ser_g:6.1f}   graph, two parallel branches {par:6.1f}")
