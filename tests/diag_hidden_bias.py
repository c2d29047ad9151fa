"""Diagnostic (not a test): which loss term explains a whole-step gradient difference on clf_t.hidden.bias.
Run on the GPU box:  python tests/diag_hidden_bias.py [L B seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import test_gpu_full_step as T
from oracle import restatement as R

L, B, seed = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 2, 1024)))
js = R.build_joint_step(L, 1, L, 1, 4, 4, seed=seed, dropout_p=0.0, zero_end=False)
tr = T._trainer_from(js, L, L, 4)
gen = torch.Generator().manual_seed(seed + 1)
batch = (T._pair(gen, B, 1, L, 4), T._pair(gen, B, 1, L, 4))
ts = (L // 8, L // 16)
# per-loss oracle gradients from a twin of the oracle (same seed -> same parameters)
js2 = R.build_joint_step(L, 1, L, 1, 4, 4, seed=seed, dropout_p=0.0, zero_end=False)
(x_t, y_t), (x_s, y_s) = batch
Ls, aux = js2.forward_losses(x_t, y_t, x_s, y_s, ts)
names = ["hidden.bias", "hidden.weight"]
per = {}
for k, v in Ls.items():
    g = torch.autograd.grad(v, [js2.m["clf_t"][n] for n in names], retain_graph=True, allow_unused=True)
    per[k] = [None if t is None else t.numpy().copy() for t in g]
rep_o, want, rep, grads = T._step_both(js, tr, batch, ts)
print("w_t", rep_o["w_t"], "w_s", rep_o["w_s"], "coeffs(a,b,c,d)", R.loss_coefficients(0))
for i, n in enumerate(names):
    got = grads["clf_t"][n].cpu().numpy().astype(np.float64); w = want["clf_t"][n].astype(np.float64)
    d = got - w
    print(f"== clf_t {n}: |want|max {np.abs(w).max():.4e}  |diff|max {np.abs(d).max():.4e}")
    if n == "hidden.bias":
        print("   got ", got, "\n   want", w, "\n   diff", d)
    for k, g in per.items():
        if g[i] is None:
            continue
        gi = g[i].astype(np.float64)
        c = float((d * gi).sum() / max((gi * gi).sum(), 1e-300))
        res = np.abs(d - c * gi).max()
        print(f"   {k:10s} |g|max {np.abs(gi).max():.3e}   best-fit coefficient {c:+.4e}   residual {res:.3e}")
scale = max(float(np.abs(v).max()) for v in want["clf_t"].values())
print("module scale", scale, "at", max(want["clf_t"], key=lambda k: float(np.abs(want["clf_t"][k]).max())))
print("logit_s2t oracle", rep_o["logit_s2t"], "\nlogit_s2t device", rep["logit_s2t"])
