"""Diagnostic: the whole joint step with the fused WN kernels vs the three-launch form, same process, same state: every
accumulated gradient (GPU vs GPU, no oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import feature_level_style_transfer_for_tsc_amd as fst
from oracle import restatement as R
from test_gpu_full_step import _pair, _trainer_from

L, B = 512, 3
js = R.build_joint_step(L, 1, L, 1, 4, 4, seed=100, dropout_p=0.0, zero_end=False)
tr = _trainer_from(js, L, L, 4)
gen = torch.Generator().manual_seed(2024)
target = _pair(gen, B, 1, L, 4)
(x_t, y_t), (x_s, y_s) = target, _pair(gen, B, 1, L, 4)
args = [v.cuda() for v in (x_t, y_t, x_s, y_s)]
snap = tr.snapshot()
res = {}
for mode in ("1", "0", "1b"):
    os.environ["FST_WN_FUSED"] = mode[0]
    tr.restore(snap)
    grads = {}
    def grab():
        for name in tr.MODULES:
            for n, p in tr.m[name].named_parameters():
                if p.grad is not None:
                    grads[f"{name}.{n}"] = p.grad.detach().clone()
    tr.on_grads_ready = grab
    rep = tr.step(*args, epoch=0, t_samples=(17, 40))
    res[mode] = (grads, {k: float(rep[k]) for k in ("nf_t", "nf_s", "cdan", "ce_s2t2s", "fd_s")})
    print(mode, res[mode][1])
for a, b in (("0", "1"), ("0", "1b"), ("1", "1b")):
    worst = {}
    for k in res[a][0]:
        x, y = res[a][0][k].double(), res[b][0][k].double()
        mod = k.split(".")[0]
        scale = max(float(v.abs().max()) for kk, v in res[a][0].items() if kk.startswith(mod + "."))
        e = float((x - y).abs().max()) / max(1e-12, scale)
        if e > worst.get(mod, (0, ""))[0]:
            worst[mod] = (e, k)
    print(f"== {a} vs {b}")
    for mod, (e, k) in worst.items():
        print(f"   {mod:13s} {e:.2e}  {k}")
