"""Parity at the shapes of the other BASELINE.json configs (the bench line is configs[1]; these are test cases):
config 0 — GunPoint-shaped classifier-only step (L=150, 2 classes, B=50);
config 3 — multivariate 9-channel, L=5000 feature extractor (LDS tiling over long sequences, forward + backward);
config 4 — L=1024 full joint forward (all nine losses) at small batch.
Each compares the HIP path with the CPU oracle on identical seeded weights and inputs (1e-4 rel on values,
1e-3 of the gradient scale on gradients)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import feature_level_style_transfer_for_tsc_amd as fst
from oracle import restatement as R

DEV = "cuda"


def close(got, want, tol, what, scale=None):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    s = scale if scale is not None else max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * s, f"{what}: max err {err:.3e}, scale {s:.3e}, tol {tol}"


def test_config0_gunpoint_shaped_classifier_step():
    gen = torch.Generator().manual_seed(150)
    fe_spec, clf_spec = R.train_specs(150, 1)
    assert len(fe_spec[0]) == 13 and fe_spec[0][-1][2] == 37            # rf = min(150//4, 89) = 37 → 13 primes
    Pf, Pc = R.init_feature_extractor(fe_spec, gen), R.init_classifier(clf_spec, 2, gen)
    tr = fst.ClassifierTrainer(150, 1, 2, DEV)
    tr.fe.load_state_dict({k: v.detach() for k, v in Pf.items()})
    tr.clf.load_state_dict({k: v.detach() for k, v in Pc.items()})
    x = torch.randn(50, 1, 150, generator=gen)
    x = (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)
    y = torch.randint(2, (50,), generator=gen)
    loss_o, logits_o = R.ClassifierStep(Pf, Pc, fe_spec, clf_spec).step(x, y)
    loss, logits = tr.step(x.to(DEV), y.to(DEV))
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    close(logits, logits_o, 1e-4, "GunPoint-shaped logits")


def test_config3_nine_channel_long_sequence_feature_extractor():
    gen = torch.Generator().manual_seed(5000)
    fe_spec, _ = R.train_specs(5000, 9)
    P = R.init_feature_extractor(fe_spec, gen)
    fe = fst.OS_CNN_res(fe_spec).to(DEV)
    fe.load_state_dict({k: v.detach() for k, v in P.items()})
    fe.train()
    x = torch.randn(2, 9, 5000, generator=gen)
    r = torch.randn(2, 50, 5000, generator=gen)
    xo = x.clone().requires_grad_(True)
    yo = R.feature_extractor(xo, P, fe_spec, True)
    (yo * r).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    y = fe(xd)
    (y * r.to(DEV)).sum().backward()
    close(y, yo, 1e-4, "FE(9ch, L=5000) forward")
    close(xd.grad, xo.grad, 1e-3, "FE(9ch, L=5000) dx")
    want = {k: v.grad for k, v in P.items() if v.grad is not None}
    scale = max(float(v.abs().max()) for v in want.values())
    for k, p in fe.named_parameters():
        close(p.grad, want[k], 1e-3, f"FE(9ch, L=5000) grad {k}", scale=scale)


def test_config4_joint_forward_losses_L1024():
    js = R.build_joint_step(1024, 1, 1024, 1, 4, 4, seed=77, dropout_p=0.0, zero_end=False)
    cfg = fst.JointConfig(L_t=1024, C_in_t=1, L_s=1024, C_in_s=1, n_class_t=4, n_class_s=4, dropout_p=0.0)
    tr = fst.JointTrainer(cfg, DEV)
    tr.load_params({k: {n: t.detach() for n, t in v.items()} for k, v in js.m.items()}, js.mats)
    gen = torch.Generator().manual_seed(1)
    mk = lambda: (torch.randn(2, 1, 1024, generator=gen), torch.randint(4, (2,), generator=gen))
    (x_t, y_t), (x_s, y_s) = mk(), mk()
    Lo, _ = js.forward_losses(x_t, y_t, x_s, y_s, (100, 37))
    Lg, _ = tr.forward_losses(x_t.to(DEV), y_t.to(DEV), x_s.to(DEV), y_s.to(DEV), (100, 37),
                              tr.m["noise"].advance(2, 2))
    for k in ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s", "cdan", "ce_s2t2s", "fd_s"):
        a, b = float(Lg[k]), float(Lo[k])
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (k, a, b)
