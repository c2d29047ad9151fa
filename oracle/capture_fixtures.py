"""Generate tests/golden/*.npz by running the REFERENCE's own modules on the CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (needs /root/reference, read-only,
which never travels to the GPU box).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_fixtures.py [--ref /root/reference]

Harness-side shims (the reference files are untouched): ``np.float`` alias,
``Tensor.cuda``/``Module.cuda`` → identity (the reference hard-codes ``.cuda()``), and a stub
``sktime`` module (DataSource.py imports it).  ``train()`` itself cannot be
driven (needs ``.ts`` datasets); the joint-step fixture drives the reference MODULES through a
harness that follows train_and_test.py:539-766 statement by statement.

Besides the module / joint-step fixtures it writes ``phases_small`` (one batch of each of the six pre-training
phase bodies of train(), from the joint fixture's initial state), ``voting_small`` (lines 281-424 of
multi_source_voting.py compiled from the reference file and executed on stub loaders / models with preset logits)
and ``datasource_small`` (the reference's TrainData / TestData classes running on top of the build's sktime-free
.ts parser, plugged in as the stub sktime's ``load_from_tsfile``).

What is stored is data only: seeds, inputs, state_dicts, outputs, gradients.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def install_shims(ref_root: str) -> None:
    np.float = float                                              # C_DAN.py:44, widgets.py:13,112
    torch.Tensor.cuda = lambda self, *a, **k: self                # OS_CNN.py:56, C_DAN.py:19 …
    torch.nn.Module.cuda = lambda self, *a, **k: self             # train_and_test.py:83-95,132
    sk, skd = types.ModuleType("sktime"), types.ModuleType("sktime.datasets")
    skd.load_from_tsfile = None
    skd.load_from_tsfile_to_dataframe = None
    sk.datasets = skd
    sys.modules["sktime"], sys.modules["sktime.datasets"] = sk, skd
    sys.path.insert(0, ref_root)


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad."):
    return {prefix + k: p.grad.detach().numpy().copy() for k, p in module.named_parameters() if p.grad is not None}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays)")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    install_shims(args.ref)
    os.makedirs(OUT, exist_ok=True)

    from OS_CNN.OS_CNN_Structure_build import generate_layer_parameter_list, get_Prime_number_in_a_range
    from OS_CNN.OS_CNN import OS_CNN_res, OS_CNN, layer_parameter_list_input_change, calculate_mask_index
    from Simplified_NF_WaveGlow import WaveGlow, WaveGlowLoss
    from C_DAN import RandomLayer, CDAN
    from widgets import (AdversarialNetworkforCDAN, FeatureDiscriminatorforSource, ProbTransfer, wgan_loss,
                         DimensionUnification, NoiseTransfer)
    from Comparison.SLARDA.train import CPC

    # ---------------------------------------------------------------- spec + mask indices
    spec_cases = []
    for (start, end, budgets, cin) in [(1, 89, [8 * 128, 5 * 128 * 256 + 2 * 256 * 128], 1),
                                       (1, 37, [8 * 128, 5 * 128 * 256 + 2 * 256 * 128], 1),
                                       (1, 89, [8 * 128 * 9, 5 * 128 * 256 + 2 * 256 * 128], 9),
                                       (1, 8, [36, 540], 1), (1, 6, [2 * 11 * 2 * 2, 700], 2)]:
        spec_cases.append({"start": start, "end": end, "budgets": budgets, "in_channel": cin,
                           "primes": get_Prime_number_in_a_range(start, end),
                           "spec": generate_layer_parameter_list(start, end, budgets, cin)})
    mask_idx = {f"{p},{kmax}": list(calculate_mask_index(p, kmax))
                for kmax in (1, 2, 3, 7, 13, 37, 89) for p in range(1, kmax + 1)}
    with open(os.path.join(OUT, "spec.json"), "w") as f:
        json.dump({"spec_cases": spec_cases, "mask_index": mask_idx}, f)
    print("wrote spec.json")

    # ---------------------------------------------------------------- FE small: fwd + bwd, train mode
    torch.manual_seed(11)
    lp = generate_layer_parameter_list(1, 6, [2 * 11 * 2 * 2, 700], 2)      # primes 1,2,3,5 (Σ=11)
    fe = OS_CNN_res(lp)
    fe.train()
    st0 = sd_np(fe, "sd.")
    x = torch.randn(4, 2, 24, requires_grad=True)
    r = torch.randn(4, sum(t[1] for t in lp[-1]), 24)
    y = fe(x)
    (y * r).sum().backward()
    save("fe_small", spec=np.array(json.dumps(lp)), x=x.detach().numpy(), r=r.numpy(), y=y.detach().numpy(),
         dx=x.grad.numpy(), **st0, **grads_np(fe), **sd_np(fe, "sd_after."))
    fe.eval()
    save("fe_small_eval", y_eval=fe(x.detach()).detach().numpy())

    # ---------------------------------------------------------------- FE metric spec (L=512, Cin=1): forward only
    torch.manual_seed(12)
    lp512 = generate_layer_parameter_list(1, 89, [8 * 128, 5 * 128 * 256 + 2 * 256 * 128], 1)
    fe512 = OS_CNN_res(lp512)
    fe512.train()
    st512 = {k: v.astype(np.float32) for k, v in sd_np(fe512, "sd.").items()}
    x512 = torch.randn(2, 1, 512)
    y512 = fe512(x512)
    save("fe_metric_fwd", x=x512.numpy(), y=y512.detach().numpy(), **st512)

    # ---------------------------------------------------------------- classifier small: train + eval
    torch.manual_seed(13)
    C = sum(t[1] for t in lp[-1])
    lp_c = layer_parameter_list_input_change(lp, C)
    clf = OS_CNN(lp_c, 3)
    clf.train()
    st0 = sd_np(clf, "sd.")
    xf = torch.randn(4, C, 24, requires_grad=True)
    logits, pooled = clf(xf)
    rl, rp = torch.randn_like(logits), torch.randn_like(pooled)
    ((logits * rl).sum() + (pooled * rp).sum()).backward()
    clf.eval()
    logits_e, pooled_e = clf(xf.detach())
    save("clf_small", spec=np.array(json.dumps(lp_c)), x=xf.detach().numpy(), rl=rl.numpy(), rp=rp.numpy(),
         logits=logits.detach().numpy(), pooled=pooled.detach().numpy(), dx=xf.grad.numpy(),
         logits_eval=logits_e.detach().numpy(), pooled_eval=pooled_e.detach().numpy(),
         **st0, **grads_np(clf), **sd_np(clf, "sd_after."))

    # ---------------------------------------------------------------- WaveGlow small
    torch.manual_seed(14)
    wg = WaveGlow(3, 6, 8)
    for k in range(3):                                            # non-degenerate coupling (end is zero-init)
        wg.WN[k].end.weight.data.normal_(0, 0.2)
        wg.WN[k].end.bias.data.normal_(0, 0.1)
    st0 = sd_np(wg, "sd.")
    xw = torch.randn(3, 6, 40, requires_grad=True)
    out = wg(xw)
    loss = WaveGlowLoss()(out)
    loss.backward()
    g0 = grads_np(wg)
    dxw = xw.grad.numpy().copy()
    zin = torch.randn(3, 6, 40, requires_grad=True)
    wg.zero_grad()
    xi = wg.infer(zin)
    ri = torch.randn_like(xi)
    (xi * ri).sum().backward()
    gi = grads_np(wg, "igrad.")
    # Q2: perturb the 1x1 weights, infer again — the cached inverse must be reused
    with torch.no_grad():
        for k in range(3):
            wg.convinv[k].conv.weight.add_(0.05 * torch.randn_like(wg.convinv[k].conv.weight))
    xi2 = wg.infer(zin.detach())
    save("waveglow_small", x=xw.detach().numpy(), z=out[0].detach().numpy(),
         log_s=np.stack([s.detach().numpy() for s in out[1]]), log_det=np.array([float(d) for d in out[2]]),
         loss=np.array(float(loss)), dx=dxw, zin=zin.detach().numpy(), xi=xi.detach().numpy(), ri=ri.numpy(),
         dzin=zin.grad.numpy(), xi2=xi2.detach().numpy(), **st0, **g0, **gi, **sd_np(wg, "sd_perturbed."))

    # ---------------------------------------------------------------- CPC small
    cpc = CPC(6, 8, 10)
    st0 = sd_np(cpc, "sd.")
    torch.manual_seed(15)
    f = torch.randn(4, 6, 20, requires_grad=True)
    torch.manual_seed(150)
    t_drawn = int(torch.randint(10 // 2, size=(1,)))
    torch.manual_seed(150)
    nce = cpc(f)
    nce.backward()
    save("cpc_small", feat=f.detach().numpy(), t_samples=np.array(t_drawn), nce=np.array(float(nce)),
         dfeat=f.grad.numpy(), **st0, **grads_np(cpc))

    # ---------------------------------------------------------------- CDAN small
    torch.manual_seed(16)
    rl_ = RandomLayer([5 * 12, 3], output_dim=32)
    ad = AdversarialNetworkforCDAN(32, 16)
    ad.dropout1.p = ad.dropout2.p = 0.0
    ad.train()
    st0 = sd_np(ad, "sd.")
    ft, fg = torch.randn(4, 5, 12, requires_grad=True), torch.randn(4, 5, 12, requires_grad=True)
    lt_, lg_ = torch.randn(4, 3, requires_grad=True), torch.randn(4, 3, requires_grad=True)
    vals, coeffs = [], []
    for _ in range(2):                                            # two calls: GRL coeff 0.9866 then ~1
        v = CDAN(ft, fg, lt_, lg_, ad, rl_)
        vals.append(float(v)); coeffs.append(float(ad.coeff))
    ad.zero_grad()
    for t in (ft, fg, lt_, lg_):
        t.grad = None
    v = CDAN(ft, fg, lt_, lg_, ad, rl_)
    v.backward()
    save("cdan_small", ft=ft.detach().numpy(), fg=fg.detach().numpy(), lt=lt_.detach().numpy(), lg=lg_.detach().numpy(),
         m0=rl_.random_matrix[0].numpy(), m1=rl_.random_matrix[1].numpy(), vals=np.array(vals + [float(v)]),
         coeffs=np.array(coeffs + [float(ad.coeff)]), dft=ft.grad.numpy(), dfg=fg.grad.numpy(),
         dlt=lt_.grad.numpy(), dlg=lg_.grad.numpy(), **st0, **grads_np(ad))

    # ---------------------------------------------------------------- small heads
    torch.manual_seed(17)
    nt = NoiseTransfer(6, 10)
    du = DimensionUnification(4, 6, 14, 10)
    pt = ProbTransfer(6)
    fd = FeatureDiscriminatorforSource(6)
    fd.train()
    heads = {}
    for name, m in (("noise", nt), ("dimunif", du), ("probtransfer", pt), ("fd_s", fd)):
        heads.update(sd_np(m, f"sd.{name}."))
    zt1, zs1 = torch.randn(3, 6, 10), torch.randn(5, 6, 10)
    zt2, zs2 = torch.randn(3, 6, 10), torch.randn(5, 6, 10, requires_grad=True)
    n1 = nt(zt1, zs1)
    n2 = nt(zt2, zs2)
    rn = torch.randn_like(n2)
    (n2 * rn).sum().backward()
    xs = torch.randn(3, 4, 14)
    pooled_in = torch.randn(5, 6, requires_grad=True)   # GRL hooks need a grad-requiring input
    fd_vals = [fd(pooled_in) for _ in range(3)]
    save("heads_small", zt1=zt1.numpy(), zs1=zs1.numpy(), zt2=zt2.numpy(), zs2=zs2.detach().numpy(),
         n1=n1.detach().numpy(), n2=n2.detach().numpy(), rn=rn.numpy(), dzs2=zs2.grad.numpy(),
         **grads_np(nt, "grad.noise."), xs=xs.numpy(), du_out=du(xs).detach().numpy(), pooled_in=pooled_in.detach().numpy(),
         pt_out=pt(pooled_in).detach().numpy(), fd_out=np.stack([v.detach().numpy() for v in fd_vals]),
         wgan=np.array(float(wgan_loss(fd_vals[0], fd_vals[1], fd_vals[2]))), **heads)

    # ---------------------------------------------------------------- 2-step joint run (train_and_test.py:539-766)
    torch.manual_seed(18)
    L_t, C_in_t, L_s, C_in_s, ncls_t, ncls_s, B = 32, 1, 24, 2, 3, 4, 4
    lp_t = generate_layer_parameter_list(1, 8, [18 * 2 * C_in_t, 10 * 18 * 3], C_in_t)
    lp_s = generate_layer_parameter_list(1, 6, [11 * 2 * C_in_s, 8 * 11 * 3], C_in_s)
    C = sum(t[1] for t in lp_t[-1])
    C_s = sum(t[1] for t in lp_s[-1])
    lp_clf = layer_parameter_list_input_change(lp_t, C)
    M = {"fe_t": OS_CNN_res(lp_t), "clf_t": OS_CNN(lp_clf, ncls_t), "fe_s": OS_CNN_res(lp_s),
         "dimunif": DimensionUnification(C_s, C, L_s, L_t), "clf_s": OS_CNN(lp_clf, ncls_s)}
    M["probtransfer"] = ProbTransfer(M["clf_s"].length_before_classification)
    M["nf"] = WaveGlow(3, C, 16)
    for k in range(3):
        M["nf"].WN[k].end.weight.data.normal_(0, 0.05)
        M["nf"].WN[k].end.bias.data.normal_(0, 0.02)
    M["noise"] = NoiseTransfer(C, L_t)
    rnd = RandomLayer([C * L_t, ncls_t], output_dim=64)
    M["ad_net"] = AdversarialNetworkforCDAN(64, 32)
    M["ad_net"].dropout1.p = M["ad_net"].dropout2.p = 0.0
    M["fd_s"] = FeatureDiscriminatorforSource(M["clf_s"].length_before_classification)
    M["cpc"] = CPC(C, 8, L_t // 2)
    nf_loss, ce = WaveGlowLoss(), torch.nn.CrossEntropyLoss()
    joint = {"meta": np.array(json.dumps({"L_t": L_t, "C_in_t": C_in_t, "L_s": L_s, "C_in_s": C_in_s, "ncls_t": ncls_t,
                                           "ncls_s": ncls_s, "B": B, "lp_t": lp_t, "lp_s": lp_s, "lp_clf": lp_clf,
                                           "nf": [3, C, 16], "cpc": [C, 8, L_t // 2]}))}
    for name, m in M.items():
        joint.update(sd_np(m, f"sd0.{name}."))
    joint["m0"], joint["m1"] = rnd.random_matrix[0].numpy(), rnd.random_matrix[1].numpy()
    rms_lr = {"fe_t": 0.001, "clf_t": 0.003, "fe_s": 0.001, "dimunif": 0.001, "clf_s": 0.003, "probtransfer": 0.001,
              "nf": 0.001, "noise": 0.005, "ad_net": 0.001, "fd_s": 0.001}
    opts = [torch.optim.RMSprop(M[k].parameters(), lr=lr) for k, lr in rms_lr.items()]
    opt_cpc = torch.optim.Adam(M["cpc"].parameters(), lr=0.002)
    w_t = torch.nn.Parameter(torch.tensor([2, 5]).float())
    w_s = torch.nn.Parameter(torch.tensor([2, 2, 4]).float())
    opt_w_t, opt_w_s = torch.optim.Adam([w_t], lr=0.0002), torch.optim.Adam([w_s], lr=0.001)
    init_t = init_s = None
    for m in M.values():
        m.train()
    import copy
    sd0_all = {name: copy.deepcopy(m.state_dict()) for name, m in M.items()}        # for the pre-training phases below
    for step in range(2):
        x_t, y_t = torch.randn(B, C_in_t, L_t), torch.randint(ncls_t, (B,))
        x_s, y_s = torch.randn(B, C_in_s, L_s), torch.randint(ncls_s, (B,))
        joint[f"s{step}.x_t"], joint[f"s{step}.y_t"] = x_t.numpy(), y_t.numpy()
        joint[f"s{step}.x_s"], joint[f"s{step}.y_s"] = x_s.numpy(), y_s.numpy()
        seeds = (500 + 2 * step, 501 + 2 * step)
        ts = []
        for s in seeds:
            torch.manual_seed(s)
            ts.append(int(torch.randint((L_t // 2) // 2, size=(1,))))
        joint[f"s{step}.t_samples"] = np.array(ts)
        feat_t = M["fe_t"](x_t)                                                     # :547
        torch.manual_seed(seeds[0])
        sl_t = M["cpc"](feat_t)                                                     # :548
        feat_s = M["dimunif"](M["fe_s"](x_s))                                       # :549-550
        torch.manual_seed(seeds[1])
        sl_s = M["cpc"](feat_s)                                                     # :551
        out_t, out_s = M["nf"](feat_t), M["nf"](feat_s)                             # :552-553
        nf_t, nf_s = nf_loss(out_t), nf_loss(out_s)
        z_s2t = M["noise"](out_t[0], out_s[0])                                      # :560
        feat_s2t = M["nf"].infer(z_s2t)                                             # :561
        logit_t, pool_t = M["clf_t"](feat_t)                                        # :583
        M["clf_t"].eval()
        logit_s2t, pool_s2t = M["clf_t"](feat_s2t)
        M["clf_t"].train()
        logit_s, pool_s = M["clf_s"](feat_s)
        ce_t, ce_s = ce(logit_t, y_t), ce(logit_s, y_s)
        cdan = CDAN(feat_t, feat_s2t, logit_t, logit_s2t, M["ad_net"], rnd)         # :593
        tr_t, tr_s2t = M["probtransfer"](pool_t), M["probtransfer"](pool_s2t)
        ce_s2t2s = ce(M["clf_s"].hidden(tr_s2t), y_s)
        fd_l = wgan_loss(M["fd_s"](tr_t), M["fd_s"](tr_s2t), M["fd_s"](pool_s))     # :601-603
        losses = {"nf_t": nf_t, "nf_s": nf_s, "ce_t": ce_t, "sl_t": sl_t, "ce_s": ce_s, "sl_s": sl_s, "cdan": cdan,
                  "ce_s2t2s": ce_s2t2s, "fd_s": fd_l}
        for k, v in losses.items():
            joint[f"s{step}.loss.{k}"] = np.array(float(v))
        joint[f"s{step}.logit_t"] = logit_t.detach().numpy()
        joint[f"s{step}.logit_s2t"] = logit_s2t.detach().numpy()
        joint[f"s{step}.feat_s2t"] = feat_s2t.detach().numpy()
        lt = torch.stack([nf_t, ce_t])
        ls = torch.stack([nf_s, ce_s, ce_s2t2s])
        if init_t is None:
            init_t = 1 / (1 + np.exp(-lt.data.cpu().numpy()))
            init_s = 1 / (1 + np.exp(-ls.data.cpu().numpy()))
        total = torch.sum(torch.mul(w_t, lt)) + torch.sum(torch.mul(w_s, ls))
        total = total + 3 * cdan + 3 * fd_l + 2 * sl_t + 2 * sl_s                  # epoch < 12 (:665-666)
        for o in opts:
            o.zero_grad()
        opt_cpc.zero_grad(); opt_w_s.zero_grad(); opt_w_t.zero_grad()
        total.backward(retain_graph=True)
        opt_w_s.zero_grad(); opt_w_t.zero_grad()
        sh_t, sh_s = M["fe_t"].return_last_layer(), M["fe_s"].return_last_layer()
        norms_t, norms_s = [], []
        for i in range(len(lt)):
            g = torch.autograd.grad(lt[i], sh_t.parameters(), retain_graph=True)
            norms_t.append(torch.cat([torch.norm(torch.mul(w_t[i], gg)).unsqueeze(0) for gg in g]).sum())
        for i in range(len(ls)):
            g = torch.autograd.grad(ls[i], sh_s.parameters(), retain_graph=True)
            norms_s.append(torch.cat([torch.norm(torch.mul(w_s[i], gg)).unsqueeze(0) for gg in g]).sum())
        nt_, ns_ = torch.stack(norms_t), torch.stack(norms_s)
        joint[f"s{step}.norms_t"], joint[f"s{step}.norms_s"] = nt_.detach().numpy(), ns_.detach().numpy()
        ratio_t = (1 / (1 + np.exp(-lt.data.cpu().numpy()))) / init_t
        ratio_s = (1 / (1 + np.exp(-ls.data.cpu().numpy()))) / init_s
        inv_t, inv_s = ratio_t / np.mean(ratio_t), ratio_s / np.mean(ratio_s)
        c_t = torch.tensor(np.mean(nt_.data.cpu().numpy()) * (inv_t ** 3), requires_grad=False)
        c_s = torch.tensor(np.mean(ns_.data.cpu().numpy()) * (inv_s ** 3), requires_grad=False)
        g_w_t = torch.autograd.grad(torch.sum(torch.abs(nt_ - c_t)), w_t)[0]
        g_w_s = torch.autograd.grad(torch.sum(torch.abs(ns_ - c_s)), w_s)[0]
        sv_t, sv_s = w_t.data.cpu().numpy(), w_s.data.cpu().numpy()
        total.data = total.data * 0.0                                               # :734-741
        w_t.data = w_t.data * 0.0
        w_s.data = w_s.data * 0.0
        lt.data = lt.data * 0.0
        ls.data = ls.data * 0.0
        cdan.data = cdan.data * 0.0
        fd_l.data = fd_l.data * 0.0
        total.backward()
        if step == 0:                                                               # accumulated grads (Q3 evidence)
            for name in M:
                joint.update(grads_np(M[name], f"s0.grad.{name}."))
        w_t.data, w_s.data = torch.tensor(sv_t), torch.tensor(sv_s)
        w_t.grad, w_s.grad = g_w_t, g_w_s
        opt_w_t.step(); opt_w_s.step()
        for o in opts:
            o.step()
        opt_cpc.step()
        w_t.data[:].clamp_(min=0.0)
        w_t.data = w_t.data * (7 / torch.sum(w_t.data, dim=0))
        w_s.data[:].clamp_(min=0.0)
        w_s.data = w_s.data * (8 / torch.sum(w_s.data, dim=0))
        for p in M["ad_net"].parameters():
            p.data.clamp_(-0.0005, 0.0005)
        for p in M["fd_s"].parameters():
            p.data.clamp_(-0.01, 0.01)
        joint[f"s{step}.w_t"], joint[f"s{step}.w_s"] = w_t.data.numpy().copy(), w_s.data.numpy().copy()
        if step == 0:                                                               # state after ONE optimiser step
            for name in M:
                joint.update(sd_np(M[name], f"sd1.{name}."))
    for k in ("sd1.fd_s.model.2.weight", "s0.grad.fd_s.model.2.weight"):      # 1.3 MB each; the other fd_s tensors pin the update
        joint.pop(k)
    save("joint_small", **joint)

    # ---------------------------------------------------------------- pre-training phases (train_and_test.py:141-494)
    # One batch of each phase body, every time from the joint fixture's initial state sd0 and on its step-0 batch, with
    # fresh optimisers as at the top of train(): losses, every accumulated gradient, and the BatchNorm running mean of
    # the target classifier's first layer after the step (phase "ssl" runs the classifiers in train mode without
    # optimising them, so their running statistics still move).
    x_t, y_t = torch.tensor(joint["s0.x_t"]), torch.tensor(joint["s0.y_t"])
    x_s, y_s = torch.tensor(joint["s0.x_s"]), torch.tensor(joint["s0.y_s"])
    seeds = (500, 501)
    phases = {"meta": np.array(json.dumps({"phases": ["target_pretrain", "source_pretrain", "ssl_with_ce", "ssl",
                                                      "nf_with_ce", "nf"], "t_samples": [int(v) for v in joint["s0.t_samples"]]}))}

    def cpc_pair(feat_t, feat_s):
        torch.manual_seed(seeds[0]); a = M["cpc"](feat_t)
        torch.manual_seed(seeds[1]); b_ = M["cpc"](feat_s)
        return a, b_

    for phase in ("target_pretrain", "source_pretrain", "ssl_with_ce", "ssl", "nf_with_ce", "nf"):
        for name, m in M.items():
            m.load_state_dict(copy.deepcopy(sd0_all[name]))
            m.train()
            for p_ in m.parameters():
                p_.grad = None
        opts = {k: torch.optim.RMSprop(M[k].parameters(), lr=lr) for k, lr in rms_lr.items()}
        opt_cpc = torch.optim.Adam(M["cpc"].parameters(), lr=0.002)
        L = {}
        if phase == "target_pretrain":                                              # :143-171
            feat_t = M["fe_t"](x_t)
            torch.manual_seed(seeds[0]); L["sl_t"] = M["cpc"](feat_t)
            L["ce_t"] = ce(M["clf_t"](feat_t)[0], y_t)
            total, stepped = L["ce_t"] + L["sl_t"], ["fe_t", "clf_t", "cpc"]
        elif phase == "source_pretrain":                                            # :183-209
            feat_s = M["dimunif"](M["fe_s"](x_s))
            L["ce_s"] = ce(M["clf_s"](feat_s)[0], y_s)
            total, stepped = L["ce_s"], ["fe_s", "dimunif", "clf_s"]
        elif phase in ("ssl_with_ce", "ssl", "nf_with_ce"):                          # :232-275, :296-348, :388-431
            feat_t = M["fe_t"](x_t)
            feat_s = M["dimunif"](M["fe_s"](x_s))
            torch.manual_seed(seeds[0]); L["sl_t"] = M["cpc"](feat_t)
            L["ce_t"] = ce(M["clf_t"](feat_t)[0], y_t)
            torch.manual_seed(seeds[1]); L["sl_s"] = M["cpc"](feat_s)
            L["ce_s"] = ce(M["clf_s"](feat_s)[0], y_s)
            if phase == "ssl_with_ce":
                total = L["sl_t"] + L["sl_s"] + 0.8 * L["ce_t"] + 1.2 * L["ce_s"]
                stepped = ["fe_t", "clf_t", "cpc", "fe_s", "dimunif", "clf_s"]
            elif phase == "ssl":
                total, stepped = L["sl_t"] + L["sl_s"], ["fe_t", "cpc", "fe_s", "dimunif"]
            else:
                L["nf_t"], L["nf_s"] = nf_loss(M["nf"](feat_t)), nf_loss(M["nf"](feat_s))
                total = L["nf_t"] + L["nf_s"] + 5 * L["ce_t"] + 5 * L["ce_s"] + 3 * L["sl_t"] + 3 * L["sl_s"]
                stepped = ["fe_t", "clf_t", "fe_s", "dimunif", "clf_s", "nf", "cpc"]
        else:                                                                       # "nf": :457-494, features detached
            feat_t = M["fe_t"](x_t).detach_()
            feat_s = M["dimunif"](M["fe_s"](x_s)).detach_()
            L["nf_t"], L["nf_s"] = nf_loss(M["nf"](feat_t)), nf_loss(M["nf"](feat_s))
            total, stepped = L["nf_t"] + L["nf_s"], ["fe_t", "fe_s", "dimunif", "nf"]
        total.backward()
        for k, v in L.items():
            phases[f"{phase}.loss.{k}"] = np.array(float(v))
        phases[f"{phase}.total"] = np.array(float(total))
        for name in ("fe_t", "clf_t", "fe_s", "dimunif", "clf_s", "nf", "cpc"):
            phases.update(grads_np(M[name], f"{phase}.grad.{name}."))
        for k in stepped:
            (opt_cpc if k == "cpc" else opts[k]).step()
        phases[f"{phase}.stepped"] = np.array(json.dumps(stepped))
        phases[f"{phase}.after.clf_t.bn_mean0"] = M["clf_t"].state_dict()["net.0.bn.running_mean"].numpy().copy()
        phases[f"{phase}.after.fe_t.w0"] = M["fe_t"].state_dict()["net_1.net.net.0.conv1d.weight"].numpy().copy()
        phases[f"{phase}.after.clf_t.hidden"] = M["clf_t"].state_dict()["hidden.weight"].numpy().copy()
    save("phases_small", **phases)

    # ---------------------------------------------------------------- multi-source voting (multi_source_voting.py:281-424)
    # The reference is a top-level script bound to its datasets and three checkpoints; its voting block (per-class
    # precision weights from train-set predictions, entropy-sharpened, weight-scaled probability sum) is executed here
    # verbatim — lines 281-424 compiled from the reference file — against stub loaders and stub "models" that return
    # preset logits, so the golden vectors are the reference's own arithmetic.
    from scipy.stats import entropy
    from sklearn.metrics import accuracy_score
    votes = {}
    with open(os.path.join(args.ref, "multi_source_voting.py"), encoding="utf-8") as f:
        lines = f.read().splitlines()
    block = compile("\n" * 280 + "\n".join(lines[280:424]), "multi_source_voting.py", "exec")
    for case, (n_class, n_train, n_test, seed, bsz) in {"a": (4, 37, 29, 1, 8), "b": (3, 50, 41, 2, 20), "c": (5, 23, 17, 3, 6)}.items():
        rng = np.random.RandomState(seed)
        y_train, y_test = rng.randint(n_class, size=n_train), rng.randint(n_class, size=n_test)
        tr_logits = rng.randn(3, n_train, n_class).astype(np.float32) * 1.5
        te_logits = rng.randn(3, n_test, n_class).astype(np.float32) * 1.5
        for k in range(3):                                  # make the models informative, differently so
            tr_logits[k, np.arange(n_train), y_train] += 0.8 * (k + 1)
            te_logits[k, np.arange(n_test), y_test] += 0.8 * (k + 1)
        if case == "c":
            tr_logits[:, :, 4] -= 50.0                      # class 4 is never predicted by any model: 0/0 -> nan -> 0
        table = {"train": tr_logits, "test": te_logits}
        state = {"which": "train"}

        def loader(which, n, y):
            def it():
                state["which"] = which
                for i0 in range(0, n, bsz):
                    idx = torch.arange(i0, min(n, i0 + bsz))
                    yield idx.double(), torch.tensor(y[i0: i0 + bsz])
            class L:                                        # enumerate(loader) restarts it, like a DataLoader
                def __iter__(self_):
                    return it()
            return L()

        def model(k):
            return lambda x: (torch.tensor(table[state["which"]][k])[x.long()], None)

        ns = {"torch": torch, "np": np, "entropy": entropy, "accuracy_score": accuracy_score, "target_num_class": n_class,
              "target_train_loader": loader("train", n_train, y_train), "target_test_loader": loader("test", n_test, y_test)}
        for k in range(3):
            ns[f"target_feature_extraction_module{k + 1}"] = lambda x: x
            ns[f"target_classification_module{k + 1}"] = model(k)
        exec(block, ns)
        votes[f"{case}.train_logits"], votes[f"{case}.test_logits"] = tr_logits, te_logits
        votes[f"{case}.train_labels"], votes[f"{case}.test_labels"] = y_train, y_test
        votes[f"{case}.weights"] = np.stack([ns["weight_1"], ns["weight_2"], ns["weight_3"]])
        votes[f"{case}.scores"] = ns["result_final"]
        votes[f"{case}.pred"] = ns["predict_list"]
        votes[f"{case}.acc"] = np.array(float(ns["acc"]))
    save("voting_small", **votes)

    # ---------------------------------------------------------------- dataset classes (DataSource.py:9-64)
    # sktime is not in the image, so the .ts PARSER has no reference vectors (restated from the format's description,
    # "parity unpinned").  The label-dictionary semantics of TrainData / TestData are pinned: the REFERENCE's classes
    # run here on top of the build's parser (plugged in as the stub sktime's load_from_tsfile).
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from feature_level_style_transfer_for_tsc_amd.data import load_ts
    sys.modules["sktime.datasets"].load_from_tsfile = lambda path, return_data_type="numpy3d": load_ts(path)
    import importlib
    import DataSource as RefDS
    importlib.reload(RefDS)
    head = "@problemName toy\n@timeStamps false\n@missing false\n@univariate false\n@dimensions 2\n@equalLength true\n" \
           "@seriesLength 4\n@classLabel true walk run sit\n@data\n"
    rng = np.random.RandomState(5)

    def ts_text(labels):
        rows = []
        for lab in labels:
            dims = [",".join(f"{v:.6g}" for v in rng.randn(4)) for _ in range(2)]
            rows.append(":".join(dims + [lab]))
        return head + "\n".join(rows) + "\n"

    texts = {"target_TRAIN": ts_text(["run", "walk", "run", "sit", "walk"]), "target_TEST": ts_text(["sit", "run", "jump", "walk"]),
             "source_TRAIN": ts_text(["sit", "sit", "hop", "run"])}
    ds = {}
    with tempfile.TemporaryDirectory() as d:
        for k, t in texts.items():
            with open(os.path.join(d, k + ".ts"), "w") as f:
                f.write(t)
            ds[f"text.{k}"] = np.array(t)
        label_dict = {}
        a = RefDS.TrainData(d, "target_TRAIN.ts", label_dict)
        ds["dict_after_target_train"] = np.array(json.dumps(label_dict))
        b = RefDS.TestData(d, "target_TEST.ts", label_dict)           # "jump" is unseen: reported, sample left unlabeled
        c = RefDS.TrainData(d, "source_TRAIN.ts", label_dict)         # shared dict: only "hop" is new -> num_class == 1
        ds["dict_final"] = np.array(json.dumps(label_dict))
        for name, obj, xk, yk in (("target_TRAIN", a, "train_x", "train_y"), ("target_TEST", b, "test_x", "test_y"),
                                  ("source_TRAIN", c, "train_x", "train_y")):
            ds[f"{name}.x"], ds[f"{name}.y"] = getattr(obj, xk).numpy(), getattr(obj, yk).numpy()
            ds[f"{name}.meta"] = np.array([obj.len, obj.in_channel, obj.time_length, obj.num_class])
            ds[f"{name}.item1_x"], ds[f"{name}.item1_y"] = obj[1][0].numpy(), np.array(int(obj[1][1]))
    save("datasource_small", **ds)


if __name__ == "__main__":
    main()
