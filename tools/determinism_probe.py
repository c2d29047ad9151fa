#!/usr/bin/env python3
"""Diagnostics: where do two runs of the joint step from the same state differ?

From one snapshot of a trainer at the bench configuration (B=256, L=512) the step is run four times — eagerly twice, through
the captured hipGraph twice — and every reported tensor (nine losses, three logit sets, feat_t, feat_s2t, GradNorm norms and
weights) plus the post-step state is compared pairwise: eager/eager and graph/graph differences are run-to-run
non-determinism (atomics, races), eager/graph differences with those two at zero are path differences (addresses, library
algorithm selection under capture).  Prints max |a-b| / max|b| per key."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import feature_level_style_transfer_for_tsc_amd as fst  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("PROBE_B", 256))
L = int(os.environ.get("PROBE_L", 512))


def pair(gen, B, L):
    x = torch.randn(B, 1, L, generator=gen)
    x = (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)
    return x, torch.randint(4, (B,), generator=gen)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(1e-300, float(b.abs().max())))


def main():
    torch.manual_seed(1234)
    tr = fst.JointTrainer(fst.JointConfig(L_t=L, L_s=L, dropout_p=0.0), DEV)
    gen = torch.Generator().manual_seed(7)
    (x_t, y_t), (x_s, y_s) = pair(gen, B, L), pair(gen, B, L)
    args = (x_t.to(DEV), y_t.to(DEV), x_s.to(DEV), y_s.to(DEV))
    tr.capture(*args, epoch=0)
    snap = tr.snapshot()
    runs = {}
    for name in ("graph1", "graph2", "eager1", "eager2"):
        tr.restore(snap)
        if name.startswith("graph"):
            rep = tr.replay(*args, (31, 77))
        else:
            rep = tr.step(*args, epoch=0, t_samples=(31, 77))
        out = {k: v.detach().clone() for k, v in rep.items()}
        after = tr.snapshot()["t"]
        for k in ("m.fe_t.net_1.net.net.1.conv1d.weight", "m.nf.WN.0.in_layers.3.weight_v", "m.nf.WN.0.in_layers.3.bias",
                  "m.nf.WN.2.res_skip_layers.7.bias", "m.clf_t.hidden.weight", "m.cpc.Wk.0.weight", "m.ad_net.ad_layer1.weight",
                  "m.fe_t.net_1.net.net.0.bn.running_var", "m.nf.WN.1.start.bias"):
            if k in after:
                out["after:" + k] = after[k].clone()
        runs[name] = out
    keys = list(runs["graph1"].keys())
    print(f"{'key':58s} {'eager/eager':>12s} {'graph/graph':>12s} {'graph/eager':>12s}   max|ref|")
    for k in keys:
        ee = rel(runs["eager1"][k], runs["eager2"][k])
        gg = rel(runs["graph1"][k], runs["graph2"][k])
        ge = rel(runs["graph1"][k], runs["eager1"][k])
        print(f"{k:58s} {ee:12.3e} {gg:12.3e} {ge:12.3e}   {float(runs['eager1'][k].abs().max()):.4e}")


if __name__ == "__main__":
    main()
