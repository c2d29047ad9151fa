"""Diagnostic: WaveGlow forward + loss + backward on realistic features (the four-source test's first pipeline), fused WN
kernels vs the three-launch form in ONE process: every saved tensor and every gradient."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import ops
from oracle import restatement as R
from test_gpu_full_step import _pair, _trainer_from

L, B = 512, 3
js = R.build_joint_step(L, 1, L, 1, 4, 4, seed=100, dropout_p=0.0, zero_end=False)
tr = _trainer_from(js, L, L, 4)
gen = torch.Generator().manual_seed(2024)
x_t, _ = _pair(gen, B, 1, L, 4)
feat = tr.m["fe_t"](x_t.cuda()).detach()
print("feat scale", float(feat.abs().max()), float(feat.std()))
res = {}
captured = {}
orig_save = None
for mode in ("0", "1"):
    os.environ["FST_WN_FUSED"] = mode
    f = feat.clone().requires_grad_(True)
    tr.m["nf"].zero_grad(set_to_none=True)
    saved = []
    def pack(t):
        saved.append(t)
        return t
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        out = tr.m["nf"](f)
    loss = fst.WaveGlowLoss()(out)
    loss.backward()
    res[mode] = {"loss": loss.detach(), "z": out[0].detach(), "df": f.grad.clone(),
                 **{"g." + n: p.grad.clone() for n, p in tr.m["nf"].named_parameters() if p.grad is not None}}
    captured[mode] = [t.detach().clone() for t in saved if isinstance(t, torch.Tensor) and t.dim() == 3]
    print(mode, "loss", float(loss), "saved 3-d tensors", len(captured[mode]))
worst = []
for k in res["0"]:
    a, b = res["0"][k].double(), res["1"][k].double()
    scale = max(1e-12, float(a.abs().max()))
    worst.append((float((a - b).abs().max()) / scale, k))
worst.sort(reverse=True)
for e, k in worst[:12]:
    print(f"{k:40s} {e:.2e}")
# gate halves of the first WN call: unfused saves g(t|s) per layer, fused saves ts per layer
ts0 = [t for t in captured["0"] if t.shape[1] == 240]
ts1 = [t for t in captured["1"] if t.shape[1] == 240]
print("ts tensors", len(ts0), len(ts1))
for i, (a, b) in enumerate(zip(ts0, ts1)):
    d = (a - b).abs()
    print(i, "max |dt,s|", float(d.max()), "mean", float(d.mean()), " max|t| ", float(a[:, :120].abs().max()))
    if i >= 9:
        break
