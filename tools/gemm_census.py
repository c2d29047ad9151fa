#!/usr/bin/env python3
"""Diagnostics: which dense products of the joint step still go to the BLAS library (aten mm / addmm / bmm), by shape.

Runs one eager step at the bench configuration under a TorchDispatchMode that logs every aten matrix product with its
operand shapes and strides; prints a table sorted by FLOPs."""
import collections
import os
import sys

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst  # noqa: E402

B = int(os.environ.get("PROBE_B", 256))
L = int(os.environ.get("PROBE_L", 512))
NAMES = ("mm", "addmm", "bmm", "baddbmm", "_addmm_activation", "linear", "matmul", "mv", "addmv")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in NAMES:
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            key = (name,) + tuple((tuple(t.shape), tuple(t.stride())) for t in ts)
            self.rows[key] += 1
        return func(*args, **(kwargs or {}))


def main():
    torch.manual_seed(0)
    tr = fst.JointTrainer(fst.JointConfig(L_t=L, L_s=L), "cuda")
    g = torch.Generator().manual_seed(1)
    def pair():
        x = torch.randn(B, 1, L, generator=g)
        return x.cuda(), torch.randint(4, (B,), generator=g).cuda()
    (x_t, y_t), (x_s, y_s) = pair(), pair()
    tr.step(x_t, y_t, x_s, y_s, epoch=0)
    with Census() as c:
        tr.step(x_t, y_t, x_s, y_s, epoch=0)
    torch.cuda.synchronize()
    def flops(key):
        shapes = [s for s, _ in key[1:]]
        if key[0] in ("mm",):
            return 2 * shapes[0][0] * shapes[0][1] * shapes[1][1]
        if key[0] in ("addmm", "_addmm_activation"):
            return 2 * shapes[1][0] * shapes[1][1] * shapes[2][1]
        if key[0] == "bmm":
            return 2 * shapes[0][0] * shapes[0][1] * shapes[0][2] * shapes[1][2]
        return 0
    for key, n in sorted(c.rows.items(), key=lambda kv: -flops(kv[0]) * kv[1]):
        print(f"{n:3d} x {key[0]:18s} {flops(key) / 1e9:8.3f} GFLOP  " + "  ".join(f"{s}/{st}" for s, st in key[1:]))


if __name__ == "__main__":
    main()
