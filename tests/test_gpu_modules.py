"""Drop-in modules on the MI355X vs (1) the committed golden fixtures produced by the reference itself and
(2) the CPU oracle on fresh seeded inputs, incl. full-size property checks.  Needs an MI355X.

Tolerances (fp32): values 1e-4 rel (north_star: logits/loss within 1e-4 rel); gradients 1e-3 of the
module's gradient scale (sum-order differences; fp32 atomics in the weight-gradient reduction).
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import ops
from oracle import restatement as R

DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def tsd(d):
    return {k: torch.tensor(v) for k, v in d.items()}


def close(got, want, tol=1e-4, what="", scale=None):
    got = got.detach().double().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64)
    want = want.detach().double().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want, dtype=np.float64)
    s = scale if scale is not None else max(1e-6, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert err <= tol * s, f"{what}: max err {err:.3e}, scale {s:.3e}, tol {tol}"


def live_masks(module):
    """{parameter name: 0/1 mask} for omni-scale conv weights whose layer computes live-tap-only weight gradients
    (the classifier: nothing reads its masked-tap gradients, see os_cnn.OS_CNN)."""
    out = {}
    for name, sub_ in module.named_modules():
        if isinstance(sub_, fst.build_layer_with_layer_parameter) and not sub_.spec.dense_dw:
            out[(name + ".conv1d.weight").lstrip(".")] = sub_.weight_mask.numpy()
    return out


def check_grads(module, want, tol=1e-3, what="", grads=None):
    scale = max(float(np.abs(v).max()) for v in want.values())
    named = {k: p.grad for k, p in module.named_parameters()} if grads is None else grads
    masks = live_masks(module)
    for k, v in want.items():
        assert named[k] is not None, f"{what}{k}: no grad"
        got = named[k].detach().cpu().numpy()
        if k in masks:                                             # compare on live taps; masked taps must be 0 or dense
            extra = got * (1 - masks[k])
            assert not np.any((extra != 0) & (np.abs(extra - v) > tol * scale)), f"{what}{k}: bad masked-tap gradient"
            got, v = got * masks[k], v * masks[k]
        close(got, v, tol, f"{what}grad {k}", scale=scale)


def spec_of(g):
    return [[tuple(t) for t in l] for l in json.loads(str(g["spec"]))]


# ------------------------------------------------------------------ golden fixtures (reference outputs)
def test_feature_extractor_golden():
    g = load("fe_small")
    fe = fst.OS_CNN_res(spec_of(g)).to(DEV)
    fe.load_state_dict(tsd(sub(g, "sd.")))
    fe.train()
    x = torch.tensor(g["x"], device=DEV, requires_grad=True)
    y = fe(x)
    close(y, g["y"], 1e-4, "fe forward")
    (y * torch.tensor(g["r"], device=DEV)).sum().backward()
    close(x.grad, g["dx"], 1e-3, "fe dx")
    check_grads(fe, sub(g, "grad."), 1e-3, "fe ")                  # dense masked-tap dW included (Q1)
    for k, v in sub(g, "sd_after.").items():
        close(fe.state_dict()[k], v, 1e-4, "fe state after " + k)
    fe.eval()
    close(fe(x.detach()), load("fe_small_eval")["y_eval"], 1e-4, "fe eval")


def test_feature_extractor_metric_spec_golden():
    g = load("fe_metric_fwd")
    spec, _ = fst.specs_for(512, 1)
    fe = fst.OS_CNN_res(spec).to(DEV)
    fe.load_state_dict(tsd(sub(g, "sd.")))
    fe.train()
    close(fe(torch.tensor(g["x"], device=DEV)), g["y"], 1e-4, "fe L=512 forward")


def test_classifier_golden():
    g = load("clf_small")
    clf = fst.OS_CNN(spec_of(g), 3).to(DEV)
    clf.load_state_dict(tsd(sub(g, "sd.")))
    clf.train()
    x = torch.tensor(g["x"], device=DEV, requires_grad=True)
    logits, pooled = clf(x)
    close(logits, g["logits"], 1e-4, "logits"); close(pooled, g["pooled"], 1e-4, "pooled")
    ((logits * torch.tensor(g["rl"], device=DEV)).sum() + (pooled * torch.tensor(g["rp"], device=DEV)).sum()).backward()
    close(x.grad, g["dx"], 1e-3, "clf dx")
    check_grads(clf, sub(g, "grad."), 1e-3, "clf ")
    clf.eval()
    le, pe = clf(x.detach())
    close(le, g["logits_eval"], 1e-4, "eval logits"); close(pe, g["pooled_eval"], 1e-4, "eval pooled")


def test_waveglow_golden():
    g = load("waveglow_small")
    wg = fst.WaveGlow(3, 6, 8).to(DEV)
    wg.load_state_dict(tsd(sub(g, "sd.")))
    x = torch.tensor(g["x"], device=DEV, requires_grad=True)
    out = wg(x)
    close(out[0], g["z"], 1e-4, "z")
    close(torch.stack(out[1]), g["log_s"], 1e-4, "log_s")
    loss = fst.WaveGlowLoss()(out)
    close(loss, g["loss"], 1e-4, "nf loss")
    loss.backward()
    close(x.grad, g["dx"], 1e-3, "wg dx")
    check_grads(wg, sub(g, "grad."), 1e-3, "wg ")
    wg.zero_grad()
    zin = torch.tensor(g["zin"], device=DEV, requires_grad=True)
    xi = wg.infer(zin)
    close(xi, g["xi"], 1e-4, "infer")
    (xi * torch.tensor(g["ri"], device=DEV)).sum().backward()
    close(zin.grad, g["dzin"], 1e-3, "infer dz")
    check_grads(wg, sub(g, "igrad."), 1e-3, "wg infer ")
    assert all(c.conv.weight.grad is None for c in wg.convinv)     # Q2: cached inverse carries no gradient
    wg.load_state_dict(tsd(sub(g, "sd_perturbed.")))               # Q2: the stale inverse is reused
    close(wg.infer(zin.detach()), g["xi2"], 1e-4, "infer with stale inverse")


def test_cpc_golden():
    g = load("cpc_small")
    cpc = fst.CPC(6, 8, 10).to(DEV)
    cpc.load_state_dict(tsd(sub(g, "sd.")))
    f = torch.tensor(g["feat"], device=DEV, requires_grad=True)
    nce = cpc(f, int(g["t_samples"]))
    close(nce, g["nce"], 1e-4, "nce")
    nce.backward()
    close(f.grad, g["dfeat"], 1e-3, "cpc dfeat")
    check_grads(cpc, sub(g, "grad."), 1e-3, "cpc ")
    torch.manual_seed(150)                                         # same global-RNG draw as the reference (Q6)
    close(cpc(f.detach()), g["nce"], 1e-4, "nce with drawn t")


def test_cdan_golden():
    g = load("cdan_small")
    ad = fst.AdversarialNetworkforCDAN(32, 16).to(DEV)
    ad.load_state_dict(tsd(sub(g, "sd.")))
    ad.dropout1.p = ad.dropout2.p = 0.0
    ad.train()
    rl = fst.RandomLayer([5 * 12, 3], output_dim=32)
    rl.random_matrix = [torch.tensor(g["m0"]), torch.tensor(g["m1"])]
    rl = rl.to(DEV)
    ts = [torch.tensor(g[k], device=DEV, requires_grad=True) for k in ("ft", "fg", "lt", "lg")]
    for i in range(3):
        for t in ts:
            t.grad = None
        ad.zero_grad()
        v = fst.CDAN(*ts, ad, rl)
        close(v, g["vals"][i], 1e-4, f"cdan call {i}")
        assert abs(ad.coeff - g["coeffs"][i]) < 1e-12
    v.backward()
    for t, k in zip(ts, ("dft", "dfg", "dlt", "dlg")):
        close(t.grad, g[k], 1e-3, k)
    check_grads(ad, sub(g, "grad."), 1e-3, "ad ")


def test_small_heads_golden():
    g = load("heads_small")
    nt = fst.NoiseTransfer(6, 10).to(DEV)
    nt.load_state_dict(tsd(sub(g, "sd.noise.")))
    d = lambda k: torch.tensor(g[k], device=DEV)
    close(nt(d("zt1"), d("zs1")), g["n1"], 1e-4, "noise 1")
    zs2 = d("zs2").requires_grad_(True)
    n2 = nt(d("zt2"), zs2)
    close(n2, g["n2"], 1e-4, "noise 2 (Q5)")
    (n2 * d("rn")).sum().backward()
    close(zs2.grad, g["dzs2"], 1e-3)
    check_grads(nt, sub(g, "grad.noise."), 1e-3, "noise ")
    du = fst.DimensionUnification(4, 6, 14, 10).to(DEV)
    du.load_state_dict(tsd(sub(g, "sd.dimunif.")))
    close(du(d("xs")), g["du_out"], 1e-4, "dimunif")


def _joint_trainer(g):
    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    cfg = fst.JointConfig(L_t=meta["L_t"], C_in_t=meta["C_in_t"], L_s=meta["L_s"], C_in_s=meta["C_in_s"],
                          n_class_t=meta["ncls_t"], n_class_s=meta["ncls_s"], nf_channels=meta["nf"][2],
                          cpc_hidden=meta["cpc"][1], cdan_dim=64, ad_hidden=32, dropout_p=0.0)
    tr = fst.JointTrainer(cfg, DEV, fe_t_spec=tup(meta["lp_t"]), clf_spec=tup(meta["lp_clf"]), fe_s_spec=tup(meta["lp_s"]))
    tr.load_params({name: tsd(sub(g, f"sd0.{name}.")) for name in tr.MODULES}, [torch.tensor(g["m0"]), torch.tensor(g["m1"])])
    return tr


def test_joint_step_golden():
    """First joint step from the reference's captured state: all nine losses, GradNorm norms and weights,
    and the accumulated gradients the reference's double backward leaves behind (Q3)."""
    g = load("joint_small")
    tr = _joint_trainer(g)
    args = [torch.tensor(g[f"s0.{k}"], device=DEV) for k in ("x_t", "y_t", "x_s", "y_s")]
    ts = tuple(int(v) for v in g["s0.t_samples"])
    grads = {}

    def capture():                                                 # grads right before the optimisers consume them
        for name in tr.MODULES:
            grads[name] = {n: p.grad.detach().clone() for n, p in tr.m[name].named_parameters() if p.grad is not None}
    tr.on_grads_ready = capture
    rep = tr.step(*args, epoch=0, t_samples=ts)
    for k in ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s", "cdan", "ce_s2t2s", "fd_s"):
        want = float(g[f"s0.loss.{k}"])
        assert abs(rep[k].item() - want) <= 1e-4 * max(1.0, abs(want)), (k, rep[k].item(), want)
    close(rep["logit_t"], g["s0.logit_t"], 1e-4, "logit_t")
    close(rep["logit_s2t"], g["s0.logit_s2t"], 1e-4, "logit_s2t")
    close(rep["feat_s2t"], g["s0.feat_s2t"], 1e-4, "feat_s2t")
    close(rep["norms_t"], g["s0.norms_t"], 1e-3, "GradNorm norms_t")
    close(rep["norms_s"], g["s0.norms_s"], 1e-3, "GradNorm norms_s")
    close(rep["w_t"], g["s0.w_t"], 1e-4, "w_t"); close(rep["w_s"], g["s0.w_s"], 1e-4, "w_s")
    for name in tr.MODULES:
        check_grads(tr.m[name], sub(g, f"s0.grad.{name}."), 2e-3, f"Q3 {name} ", grads=grads[name])


@pytest.mark.parametrize("phase", ["target_pretrain", "source_pretrain", "ssl_with_ce", "ssl", "nf_with_ce", "nf"])
def test_pretraining_phase_steps_golden(phase):
    """One batch of every pre-training phase of the reference's train() (train_and_test.py:141-494) from the joint
    fixture's state: losses, which modules receive gradients and their values, the target classifier's BatchNorm
    running mean afterwards (phase "ssl" moves it without training the classifier), which optimisers moved."""
    g, ph = load("joint_small"), load("phases_small")
    tr = _joint_trainer(g)
    args = [torch.tensor(g[f"s0.{k}"], device=DEV) for k in ("x_t", "y_t", "x_s", "y_s")]
    ts = tuple(int(v) for v in g["s0.t_samples"])
    assert list(tr.PHASES[phase]) == json.loads(str(ph[f"{phase}.stepped"]))
    grads = {}

    def capture():
        for name in tr.MODULES:
            grads[name] = {n: p.grad.detach().clone() for n, p in tr.m[name].named_parameters() if p.grad is not None}
    tr.on_grads_ready = capture
    hidden0 = tr.m["clf_t"].hidden.weight.detach().clone()
    nf0 = tr.m["nf"].WN[0].end.weight.detach().clone()
    rep = tr.phase_step(phase, *args, t_samples=ts)
    want_total = float(ph[f"{phase}.total"])
    assert abs(rep["total"].item() - want_total) <= 1e-4 * max(1.0, abs(want_total)), (rep["total"].item(), want_total)
    for k in ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s"):
        if f"{phase}.loss.{k}" in ph:
            want = float(ph[f"{phase}.loss.{k}"])
            assert abs(rep[k].item() - want) <= 1e-4 * max(1.0, abs(want)), (k, rep[k].item(), want)
        else:
            assert k not in rep
    for name in ("fe_t", "clf_t", "fe_s", "dimunif", "clf_s", "nf", "cpc"):
        want = sub(ph, f"{phase}.grad.{name}.")
        assert set(grads[name]) == set(want), (phase, name, set(grads[name]) ^ set(want))
        if want:
            check_grads(tr.m[name], want, 2e-3, f"{phase} {name} ", grads=grads[name])
    close(tr.m["clf_t"].state_dict()["net.0.bn.running_mean"], ph[f"{phase}.after.clf_t.bn_mean0"], 1e-4, "clf_t BN running mean")
    assert (not torch.equal(tr.m["clf_t"].hidden.weight.detach(), hidden0)) == ("clf_t" in tr.PHASES[phase])
    assert (not torch.equal(tr.m["nf"].WN[0].end.weight.detach(), nf0)) == ("nf" in tr.PHASES[phase])


def test_hipgraph_replay_matches_eager_step():
    """The captured step (one hipGraph replay) must equal the eager step from the same state, and its
    per-replay inputs (batch, CPC start indices) must be live."""
    g = load("joint_small")
    tr = _joint_trainer(g)
    args = [torch.tensor(g[f"s0.{k}"], device=DEV) for k in ("x_t", "y_t", "x_s", "y_s")]
    torch.manual_seed(5)
    tr.capture(*args, epoch=0)
    snap = tr.snapshot()
    rep = {k: v.clone() for k, v in tr.replay(*args, (3, 5)).items()}
    after_graph = tr.snapshot()
    tr.restore(snap)
    eager = tr.step(*args, epoch=0, t_samples=(3, 5))
    for k in ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s", "cdan", "ce_s2t2s", "fd_s"):
        assert abs(rep[k].item() - eager[k].item()) <= 1e-4 * max(1.0, abs(eager[k].item())), (k, rep[k].item(), eager[k].item())
    close(rep["logit_s2t"], eager["logit_s2t"], 1e-4, "graph vs eager logit_s2t")
    close(rep["w_t"], eager["w_t"], 1e-4, "graph vs eager w_t")
    after_eager = tr.snapshot()
    for k in ("m.fe_t.net_1.net.net.1.conv1d.weight", "m.nf.WN.0.in_layers.3.weight_v", "m.cpc.Wk.0.weight", "w_s"):
        close(after_graph["t"][k], after_eager["t"][k], 2e-3, "post-step " + k)
    other = tr.replay(*args, (1, 2))
    assert abs(other["sl_t"].item() - rep["sl_t"].item()) > 1e-6      # a different CPC start index is really used


# ------------------------------------------------------------------ oracle on fresh inputs, metric shapes
def test_waveglow_metric_width_vs_oracle():
    """WaveGlow(3, 50, 120) — the real widths — forward loss, backward and infer vs the CPU oracle (B=2, L=512)."""
    gen = torch.Generator().manual_seed(21)
    P = R.init_waveglow(3, 50, 120, gen, zero_end=False)
    wg = fst.WaveGlow(3, 50, 120).to(DEV)
    wg.load_state_dict({k: v.detach() for k, v in P.items()})
    x = torch.randn(2, 50, 512, generator=gen)
    xo = x.clone().requires_grad_(True)
    out_o = R.waveglow_forward(xo, P, 3)
    loss_o = R.waveglow_loss(out_o)
    loss_o.backward()
    xd = x.to(DEV).requires_grad_(True)
    out = wg(xd)
    loss = fst.WaveGlowLoss()(out)
    loss.backward()
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    close(out[0], out_o[0], 1e-4, "z")
    close(xd.grad, xo.grad, 1e-3, "dx")
    check_grads(wg, {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}, 1e-3, "wg120 ")
    z = torch.randn(2, 50, 512, generator=gen)
    close(wg.infer(z.to(DEV)), R.waveglow_infer(z, P, 3, {}), 1e-4, "infer")


def test_metric_shape_classifier_step_vs_oracle():
    """S1 at the metric spec (L=512, C_in=1), B=8: loss, logits and post-step weights vs the oracle."""
    gen = torch.Generator().manual_seed(33)
    fe_spec, clf_spec = R.train_specs(512, 1)
    Pf, Pc = R.init_feature_extractor(fe_spec, gen), R.init_classifier(clf_spec, 4, gen)
    tr = fst.ClassifierTrainer(512, 1, 4, DEV)
    tr.fe.load_state_dict({k: v.detach() for k, v in Pf.items()})
    tr.clf.load_state_dict({k: v.detach() for k, v in Pc.items()})
    x, y = torch.randn(8, 1, 512, generator=gen), torch.randint(4, (8,), generator=gen)
    oracle = R.ClassifierStep(Pf, Pc, fe_spec, clf_spec)
    loss_o, logits_o = oracle.step(x, y)
    loss, logits = tr.step(x.to(DEV), y.to(DEV))
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    close(logits, logits_o, 1e-4, "logits")
    close(tr.clf.hidden.weight, Pc["hidden.weight"], 1e-3, "hidden.weight after step")


# ------------------------------------------------------------------ full-size properties (B=256, L=512)
def test_full_size_flow_round_trip_and_bn_moments():
    """At BASELINE's size no oracle is affordable; use properties: infer(forward(x)) = x when the cached
    inverse is fresh, and train-mode BN output has zero mean / unit variance per channel."""
    torch.manual_seed(0)
    B, C, L = 256, 50, 512
    wg = fst.WaveGlow(3, C, 120).to(DEV)
    for wn in wg.WN:
        wn.end.weight.data.normal_(0, 0.02); wn.end.bias.data.normal_(0, 0.02)
    x = torch.randn(B, C, L, device=DEV)
    with torch.no_grad():
        z, log_s, log_det = wg(x)
        back = wg.infer(z)
    close(back, x, 1e-3, "flow round trip")
    fe_spec, _ = fst.specs_for(L, 1)
    fe = fst.OS_CNN_res(fe_spec).to(DEV)
    fe.train()
    layer = fe.net_1.net.layer_list[1]
    h = fe.net_1.net.layer_list[0](torch.randn(B, 1, L, device=DEV))
    y = layer.conv(h)
    out = ops.BNActFn.apply(y, layer.bn.weight, layer.bn.bias, layer.bn.running_mean, layer.bn.running_var, True, False,
                            1e-5, 0.1)
    m, v = out.mean(dim=(0, 2)), out.var(dim=(0, 2), unbiased=False)
    assert float(m.abs().max()) < 1e-4 and float((v - 1).abs().max()) < 1e-3
    # linearity of the omni-scale conv in its input (bias removed)
    y2 = layer.conv(2.5 * h)
    b = layer.conv1d.bias.view(1, -1, 1)
    close(y2 - b, 2.5 * (y - b), 1e-4, "conv linearity")


def test_checkpoint_load_and_eval_pass_match_the_oracle(tmp_path):
    """A checkpoint in the reference's .tar format (state_dicts captured from the reference) is loaded into fresh
    modules; the on-device eval pass (utils.py:27-183) then predicts what the oracle's eval-mode forward predicts,
    target side and source side (with DimensionUnification)."""
    g = load("joint_small")
    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    pt, ps = str(tmp_path / "epoch_3.tar"), str(tmp_path / "epoch_3_source.tar")
    torch.save({"epoch": 3, "feature_extraction_state_dict": tsd(sub(g, "sd0.fe_t.")),
                "classification_state_dict": tsd(sub(g, "sd0.clf_t."))}, pt)
    torch.save({"epoch": 3, "feature_extraction_state_dict": tsd(sub(g, "sd0.fe_s.")),
                "source_to_target_feature_trans": tsd(sub(g, "sd0.dimunif.")),
                "classification_state_dict": tsd(sub(g, "sd0.clf_s."))}, ps)
    C, C_s = sum(t[1] for t in meta["lp_t"][-1]), sum(t[1] for t in meta["lp_s"][-1])
    fe_t, clf_t = fst.OS_CNN_res(tup(meta["lp_t"])).to(DEV), fst.OS_CNN(tup(meta["lp_clf"]), meta["ncls_t"]).to(DEV)
    fe_s, clf_s = fst.OS_CNN_res(tup(meta["lp_s"])).to(DEV), fst.OS_CNN(tup(meta["lp_clf"]), meta["ncls_s"]).to(DEV)
    du = fst.DimensionUnification(C_s, C, meta["L_s"], meta["L_t"]).to(DEV)
    assert fst.load_target_classification_modules(pt, fe_t, clf_t) == 3
    assert fst.load_source_classification_modules(ps, fe_s, du, clf_s) == 3
    for m in (fe_t, clf_t, fe_s, clf_s, du):
        m.eval()
    gen = torch.Generator().manual_seed(11)
    bt = [(torch.randn(5, meta["C_in_t"], meta["L_t"], generator=gen), torch.randint(meta["ncls_t"], (5,), generator=gen)) for _ in range(3)]
    bs = [(torch.randn(4, meta["C_in_s"], meta["L_s"], generator=gen), torch.randint(meta["ncls_s"], (4,), generator=gen)) for _ in range(3)]
    acc_t, pred_t = fst.eval_accuracy(fe_t, clf_t, bt)
    acc_s, pred_s = fst.eval_accuracy(fe_s, clf_s, bs, feature_trans=du)
    P = {k: {n: torch.tensor(v) for n, v in sub(g, f"sd0.{k}.").items()} for k in ("fe_t", "clf_t", "fe_s", "dimunif", "clf_s")}
    want_t, want_s, hit_t, hit_s = [], [], 0, 0
    for x, y in bt:
        logits = R.classifier(R.feature_extractor(x, P["fe_t"], tup(meta["lp_t"]), False), P["clf_t"], tup(meta["lp_clf"]), False)[0]
        want_t.append(logits.argmax(1)); hit_t += int((logits.argmax(1) == y).sum())
    for x, y in bs:
        f = R.dimension_unification(R.feature_extractor(x, P["fe_s"], tup(meta["lp_s"]), False), P["dimunif"])
        logits = R.classifier(f, P["clf_s"], tup(meta["lp_clf"]), False)[0]
        want_s.append(logits.argmax(1)); hit_s += int((logits.argmax(1) == y).sum())
    assert torch.equal(pred_t.cpu(), torch.cat(want_t)) and abs(acc_t - hit_t / 15) < 1e-9
    assert torch.equal(pred_s.cpu(), torch.cat(want_s)) and abs(acc_s - hit_s / 12) < 1e-9


def test_multi_source_voting_on_device_matches_the_oracle():
    """K = 3 target-side models (the joint fixture's extractor/classifier and two perturbed copies) vote on a test
    set: K-way eval forward + batched vote on the device == oracle eval forward + the restated voting block."""
    g = load("joint_small")
    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    gen = torch.Generator().manual_seed(21)
    models, params = [], []
    for k in range(3):
        Pf, Pc = tsd(sub(g, "sd0.fe_t.")), tsd(sub(g, "sd0.clf_t."))
        if k:
            Pc["hidden.weight"] = Pc["hidden.weight"] + 0.5 * k * torch.randn(Pc["hidden.weight"].shape, generator=gen)
            Pc["hidden.bias"] = Pc["hidden.bias"] + 0.2 * k * torch.randn(Pc["hidden.bias"].shape, generator=gen)
        fe, clf = fst.OS_CNN_res(tup(meta["lp_t"])).to(DEV), fst.OS_CNN(tup(meta["lp_clf"]), meta["ncls_t"]).to(DEV)
        fe.load_state_dict(Pf); clf.load_state_dict(Pc)
        fe.eval(); clf.eval()
        models.append((fe, clf)); params.append((Pf, Pc))
    mk = lambda n: [(torch.randn(6, meta["C_in_t"], meta["L_t"], generator=gen), torch.randint(meta["ncls_t"], (6,), generator=gen))
                    for _ in range(n)]
    train, test = mk(4), mk(3)
    w, scores, pred, acc = fst.multi_source_voting(models, train, test)

    def oracle_logits(batches):
        out = [torch.cat([R.classifier(R.feature_extractor(x, Pf, tup(meta["lp_t"]), False), Pc, tup(meta["lp_clf"]), False)[0]
                          for x, _ in batches]) for Pf, Pc in params]
        return torch.stack(out).detach().numpy(), torch.cat([y for _, y in batches]).numpy()
    (trl, try_), (tel, tey) = oracle_logits(train), oracle_logits(test)
    w_o, scores_o, pred_o, acc_o = R.multi_source_vote(trl, try_, tel, tey)
    np.testing.assert_allclose(w.cpu().numpy(), w_o, rtol=1e-9, atol=0)
    np.testing.assert_allclose(scores.cpu().numpy(), scores_o, rtol=2e-3)          # 9^w·(1+120e^-H) amplifies the 1e-4 logit tolerance
    assert np.array_equal(pred.cpu().numpy(), pred_o) and abs(acc - acc_o) < 1e-12


def test_device_loader_pinned_async_copies():
    """The H2D side of the input pipeline: float32 cast, pinned staging, one batch ahead on a side stream."""
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(37, 3, 64, generator=gen, dtype=torch.float64)
    y = torch.randint(5, (37,), generator=gen)
    ld = fst.DeviceLoader(x, y, 8, DEV)
    xs, ys = [], []
    for xb, yb in ld:
        assert xb.is_cuda and xb.dtype == torch.float32 and yb.is_cuda
        xs.append((xb * 2).cpu() / 2); ys.append(yb.cpu())            # consume on the current stream
    assert torch.equal(torch.cat(xs), x.float()) and torch.equal(torch.cat(ys), y) and len(ld) == 5
    sh = fst.DeviceLoader(x, y, 8, DEV, generator=torch.Generator().manual_seed(1), drop_last=True)
    n = 0
    for xb, yb in sh:                                                 # shuffled: every batch is a gather of the source rows
        assert xb.shape == (8, 3, 64) and yb.shape == (8,)
        n += 8
    assert n == 32 and len(sh) == 4


def test_inference_mode_folds_batchnorm_into_the_convs():
    """Eval mode with autograd off (the eval pass and the K-way voting forward): every conv → BatchNorm (→ residual add
    → ReLU) runs as ONE launch with the normalisation folded into weights and bias.  Same logits as the unfolded eval
    path (autograd on) and as the oracle; a third of the conv-engine + BatchNorm launches."""
    g = load("joint_small")
    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    fe, clf = fst.OS_CNN_res(tup(meta["lp_t"])).to(DEV), fst.OS_CNN(tup(meta["lp_clf"]), meta["ncls_t"]).to(DEV)
    Pf, Pc = tsd(sub(g, "sd0.fe_t.")), tsd(sub(g, "sd0.clf_t."))
    gen = torch.Generator().manual_seed(5)
    for P in (Pf, Pc):                                                    # non-trivial running statistics
        for k in P:
            if k.endswith("running_mean"):
                P[k] = torch.randn(P[k].shape, generator=gen) * 0.3
            if k.endswith("running_var"):
                P[k] = torch.rand(P[k].shape, generator=gen) + 0.5
    fe.load_state_dict(Pf); clf.load_state_dict(Pc)
    fe.eval(); clf.eval()
    x = torch.randn(6, meta["C_in_t"], meta["L_t"], generator=gen)
    logits_unfolded, pooled_unfolded = clf(fe(x.to(DEV)))                  # autograd on: BatchNorm kernels in eval mode
    with torch.no_grad():
        logits, pooled = clf(fe(x.to(DEV)))                                # folded
        again, _ = clf(fe(x.to(DEV)))                                      # cached fold
    want, want_pooled = R.classifier(R.feature_extractor(x, Pf, tup(meta["lp_t"]), False), Pc, tup(meta["lp_clf"]), False)
    close(logits, want, 1e-4, "folded logits vs oracle"); close(pooled, want_pooled, 1e-4, "folded pooled vs oracle")
    close(logits, logits_unfolded, 2e-5, "folded vs unfolded eval path")
    assert torch.equal(again, logits)
    # the fold follows the parameters: change a running statistic in place -> the next inference call sees it
    with torch.no_grad():
        clf.layer_list[0].bn.running_mean.add_(0.25)
        moved, _ = clf(fe(x.to(DEV)))
    Pc2 = dict(Pc); Pc2["net.0.bn.running_mean"] = Pc["net.0.bn.running_mean"] + 0.25
    want2, _ = R.classifier(R.feature_extractor(x, Pf, tup(meta["lp_t"]), False), Pc2, tup(meta["lp_clf"]), False)
    close(moved, want2, 1e-4, "logits after moving a running mean")
    # launches: timer keys count this library's GEMM-class launches; folded = 7 convs (3 + shortcut + 3) and the classifier's
    # Linear head (fst_gemm in split-bf16 mode, the library GEMM under FST_MATH=f32), nothing else
    timer = ops.KernelTimer()
    ops.KERNEL_TIMER = timer
    with torch.no_grad():
        clf(fe(x.to(DEV)))
    ops.KERNEL_TIMER = None
    counts = {k: v["launches"] for k, v in timer.summary().items()}
    assert sum(n for k, n in counts.items() if not k.startswith("gemm_bf3")) == 7, counts
    assert sum(n for k, n in counts.items() if k.startswith("gemm_bf3")) == (1 if ops.MATH == "bf16x3" else 0), counts


def test_trainer_state_checkpoint_resume_equals_uninterrupted(tmp_path):
    """JointTrainer.save_state / load_state: two steps, save, a third step — against a FRESH trainer (other initial
    weights, no optimiser state yet) that loads the file and takes the same third step.  Everything the step depends
    on must have travelled: optimiser moments, GradNorm weights and reference losses, NoiseTransfer's sums (Q5), the
    GRL counters (Q7), the random matrices, the cached stale inverses of the flow (Q2)."""
    g = load("joint_small")
    tr = _joint_trainer(g)
    args = [torch.tensor(g[f"s0.{k}"], device=DEV) for k in ("x_t", "y_t", "x_s", "y_s")]
    for i in range(2):
        tr.step(*args, epoch=0, t_samples=(2 + i, 5 - i))
    path = str(tmp_path / "trainer_state.pt")
    tr.save_state(path)
    want = tr.step(*args, epoch=0, t_samples=(4, 1))
    want_params = {k: v.detach().clone() for k, v in tr._state_tensors().items() if k.startswith("m.")}

    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    torch.manual_seed(999)                                               # a different initialisation: all of it must be replaced
    cfg = fst.JointConfig(L_t=meta["L_t"], C_in_t=meta["C_in_t"], L_s=meta["L_s"], C_in_s=meta["C_in_s"],
                          n_class_t=meta["ncls_t"], n_class_s=meta["ncls_s"], nf_channels=meta["nf"][2],
                          cpc_hidden=meta["cpc"][1], cdan_dim=64, ad_hidden=32, dropout_p=0.0)
    fresh = fst.JointTrainer(cfg, DEV, fe_t_spec=tup(meta["lp_t"]), clf_spec=tup(meta["lp_clf"]), fe_s_spec=tup(meta["lp_s"]))
    fresh.load_state(path)
    got = fresh.step(*args, epoch=0, t_samples=(4, 1))
    for k in ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s", "cdan", "ce_s2t2s", "fd_s"):
        assert abs(got[k].item() - want[k].item()) <= 2e-5 * max(1.0, abs(want[k].item())), (k, got[k].item(), want[k].item())
    close(got["w_t"], want["w_t"], 1e-5, "w_t after resume"); close(got["w_s"], want["w_s"], 1e-5, "w_s after resume")
    close(got["logit_s2t"], want["logit_s2t"], 1e-4, "logit_s2t after resume")
    got_params = fresh._state_tensors()
    for k in ("m.nf.WN.0.in_layers.3.weight_v", "m.clf_t.hidden.weight", "m.ad_net.ad_layer1.weight", "m.cpc.Wk.0.weight",
              "m.noise.apply_learnable_weight.weight"):
        # ad_net's weights are clamped to +-5e-4: its scale is tiny and run-to-run rounding (fp32 atomics) shows at 1e-6 absolute
        close(got_params[k], want_params[k], 1e-2 if ".ad_net." in k else 2e-3, "post-step " + k)
    assert fresh.m["noise"].time == tr.m["noise"].time and fresh.m["ad_net"].iter_num == tr.m["ad_net"].iter_num


def test_folded_inference_follows_graph_replays():
    """Round-2 advisor finding: the BatchNorm fold must not be cached across calls on tensor version counters — graph
    replays and fst_bn_finalize move parameters and running statistics without bumping them.  Capture, eval (folded),
    replay three steps, eval again: the second folded eval must equal the unfolded eval-mode path (autograd on, BatchNorm
    kernels reading the live running statistics) and must differ from the first."""
    g = load("joint_small")
    tr = _joint_trainer(g)
    args = [torch.tensor(g[f"s0.{k}"], device=DEV) for k in ("x_t", "y_t", "x_s", "y_s")]
    torch.manual_seed(5)
    tr.capture(*args, epoch=0)
    fe, clf = tr.m["fe_t"], tr.m["clf_t"]
    x = args[0]

    def evals():
        fe.eval(); clf.eval()
        with torch.no_grad():
            folded, _ = clf(fe(x))
            acc, _ = fst.eval_accuracy(fe, clf, [(x, args[1])])              # the eval pass itself (fold scoped to its pack_cache)
        unfolded, _ = clf(fe(x))                                              # autograd on: eval-mode BatchNorm kernels
        fe.train(); clf.train()
        return folded.clone(), unfolded.detach().clone(), acc
    f0, u0, _ = evals()
    close(f0, u0, 2e-5, "folded vs unfolded before the replays")
    for i in range(3):
        tr.replay(*args, (1 + i, 4 - i))
    f1, u1, _ = evals()
    close(f1, u1, 2e-5, "folded vs unfolded after three graph replays")
    assert float((u1 - u0).abs().max()) > 1e-4 * float(u0.abs().max()), "the replays did not move the eval logits: the test is vacuous"


def test_trainer_state_in_memory_rollback():
    """state_dict() must not alias live optimiser moments: snapshot, two more steps, load_state_dict(snapshot), one step ==
    the same step taken right after the snapshot."""
    g = load("joint_small")
    tr = _joint_trainer(g)
    args = [torch.tensor(g[f"s0.{k}"], device=DEV) for k in ("x_t", "y_t", "x_s", "y_s")]
    tr.step(*args, epoch=0, t_samples=(2, 5))
    sd = tr.state_dict()
    assert all(not v.is_cuda for st in sd["opts"]["nf"]["state"].values() for v in st.values() if isinstance(v, torch.Tensor))
    want = tr.step(*args, epoch=0, t_samples=(3, 4))
    want_w = tr.m["nf"].WN[0].in_layers[3].weight_v.detach().clone()
    tr.step(*args, epoch=0, t_samples=(1, 1))
    tr.load_state_dict(sd)
    got = tr.step(*args, epoch=0, t_samples=(3, 4))
    for k in ("nf_t", "ce_t", "sl_t", "cdan", "fd_s"):
        assert abs(got[k].item() - want[k].item()) <= 2e-5 * max(1.0, abs(want[k].item())), (k, got[k].item(), want[k].item())
    close(tr.m["nf"].WN[0].in_layers[3].weight_v, want_w, 2e-3, "weight after rollback + step (RMSprop moments restored)")


@pytest.mark.parametrize("n_layers,kernel,fused_expected", [(8, 5, False), (9, 3, True), (3, 3, True)])
def test_wn_other_depths_and_kernel_sizes_vs_oracle(n_layers, kernel, fused_expected, monkeypatch):
    """The reference's WN takes any n_layers / kernel_size (Simplified_NF_WaveGlow.py:55-99).  kernel_size = 5 must take the
    generic conv-engine path (the fused kernels are 3-tap); nine layers reach dilation 256, whose data-gradient window does
    not fit the fused data-gradient kernel's LDS ring: that layer alone falls back.  Output and every gradient vs the oracle."""
    from feature_level_style_transfer_for_tsc_amd.waveglow import WN
    h, n, B, L = 6, 16, 3, 640
    torch.manual_seed(n_layers * 10 + kernel)
    wn = WN(h, n_layers, n, kernel).to(DEV)
    wn.end.weight.data.normal_(0, 0.3); wn.end.bias.data.normal_(0, 0.3)
    assert ops.wn_dgrad_ok(n, h, 128) and not ops.wn_dgrad_ok(n, h, 256)
    P = {("WN.0." + k): v.detach().cpu().clone().requires_grad_(True) for k, v in wn.state_dict().items()}
    monkeypatch.setattr(R, "WN_LAYERS", n_layers)
    monkeypatch.setattr(R, "WN_KERNEL", kernel)
    u0 = torch.randn(B, h, L)
    r = torch.randn(B, 2 * h, L)
    uo = u0.clone().requires_grad_(True)
    (R.wn_forward(uo, P, "WN.0.") * r).sum().backward()
    want = R.wn_forward(u0, P, "WN.0.").detach()
    ud = u0.to(DEV).requires_grad_(True)
    timer = ops.KernelTimer()
    ops.KERNEL_TIMER = timer
    try:
        out = wn(ud)
        (out * r.to(DEV)).sum().backward()
    finally:
        ops.KERNEL_TIMER = None
    keys = timer.summary()
    # (the fused kernels exist in the split-bf16 arithmetic only: under FST_MATH=f32 every depth takes the conv engine)
    assert ("wn_layer_fwd_kernel" in keys) == (fused_expected and ops.MATH == "bf16x3"), sorted(keys)
    close(out, want, 1e-4, "WN output")
    close(ud.grad, uo.grad, 1e-3, "WN d input")
    check_grads(wn, {k[5:]: v.grad.numpy() for k, v in P.items() if v.grad is not None}, 1e-3, f"WN({n_layers} layers, k={kernel}) ")


def test_dimension_unification_fused_relus_vs_composition():
    """DimensionUnification with both ReLUs in the GEMM / conv epilogues (ops.LinearActFn, ops.ConvReluFn) against the reference's
    composition relu(conv1x1(relu(linear(x)))) (widgets.py:66-78) in fp64: output and every gradient."""
    torch.manual_seed(5)
    du = fst.DimensionUnification(25, 50, 96, 64).to(DEV)
    x = torch.randn(9, 25, 96, device=DEV, requires_grad=True)
    cot = torch.randn(9, 50, 64, device=DEV)
    y = du(x)
    got = torch.autograd.grad(y, [x] + list(du.parameters()), cot)
    P = {k: v.detach().double().requires_grad_(True) for k, v in du.named_parameters()}
    x64 = x.detach().double().requires_grad_(True)
    h = torch.relu(F.linear(x64, P["length_unification.weight"], P["length_unification.bias"]))
    want = torch.relu(F.conv1d(h, P["channel_unification.weight"], P["channel_unification.bias"]))
    wg = torch.autograd.grad(want, [x64] + [P[k] for k, _ in du.named_parameters()], cot.double())
    close(y, want.detach().cpu().numpy(), 1e-5, "dimunif out")
    for a, b, name in zip(got, wg, ["dx"] + [k for k, _ in du.named_parameters()]):
        close(a, b.detach().cpu().numpy(), 1e-4, name)


def test_conv_bias_gradient_comes_from_the_batchnorm_backward_launch():
    """The bias gradient of a conv that feeds a BatchNorm (OS_CNN.py:67-72) is Σ_{b,t} of the BatchNorm's input gradient:
    fst_bn_bwd_apply leaves the per-(sample, channel) sums, ConvFn.backward adds them over the batch — no row-sum pass of its
    own over dx (ops.row_sum is not called), same value as that pass; a cotangent that is NOT the BatchNorm's output (here: scaled
    by an op in between) falls back to the conv's own reduction."""
    torch.manual_seed(3)
    g = load("joint_small")
    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    fe = fst.OS_CNN_res(tup(meta["lp_t"])).to(DEV)
    fe.train()
    x = torch.randn(6, meta["C_in_t"], meta["L_t"], device=DEV)
    calls, row_sum0 = [], ops.row_sum
    ops.row_sum = lambda *a, **k: (calls.append(1), row_sum0(*a, **k))[1]
    try:
        fe(x).square().sum().backward()
    finally:
        ops.row_sum = row_sum0
    assert not calls, f"{len(calls)} row-sum launches: the BatchNorm backward's sums did not reach the convs"
    fused = {k: v.grad.clone() for k, v in fe.named_parameters() if k.endswith("conv1d.bias")}
    assert fused
    # the same gradients with the hand-over disabled: every conv reduces its own cotangent
    attach0 = ops._ROW_SUMS.attach
    summed = []                                            # Σ|dx| per channel: the scale of what each bias gradient sums
    ops._RowSums.attach = staticmethod(lambda dx, rs: summed.append(float(dx.abs().sum(dim=(0, 2)).max())))
    try:
        fe.zero_grad()
        fe(x).square().sum().backward()
    finally:
        ops._RowSums.attach = staticmethod(attach0)
    for k, v in fe.named_parameters():
        if k in fused:
            # both are sums of the same dx in different orders; a bias in front of a BatchNorm has a zero gradient in exact
            # arithmetic, so compare against the scale of what is summed, not of the (rounding-level) result
            err = float((fused[k] - v.grad).abs().max())
            assert err <= 2e-6 * max(summed), (k, err, max(summed))
    # a rewritten cotangent must not use stale sums
    dx = torch.randn(4, 5, 32, device=DEV)
    ops._ROW_SUMS.attach(dx, torch.zeros(4, 5, device=DEV))
    assert ops._ROW_SUMS.take(dx) is not None
    dx.mul_(2.0)
    assert ops._ROW_SUMS.take(dx) is None and ops._ROW_SUMS.take(dx * 1.0) is None
