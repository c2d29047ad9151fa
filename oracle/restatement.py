"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A from-the-spec CPU restatement (PyTorch CPU ops, fp32) of the train-step hot path of
BaeHann/feature_level_style_transfer_for_TSC.  It exists to CHECK the HIP path; nothing
in ``feature_level_style_transfer_for_tsc_amd/`` may import it.  Allowed importers:
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.

Pinning: the reference has no tests or golden vectors.  This restatement is
pinned against outputs of the reference itself, imported in the build container by
``oracle/capture_fixtures.py`` (harness-side shims only) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` asserts agreement.

Style: functional.  Every function takes a flat ``dict`` ``P`` mapping the reference's
``state_dict`` key names to tensors, so captured reference weights load without any module
hierarchy.  Non-persisted reference state (weight masks, cached ``W_inverse``, NoiseTransfer
accumulators, GRL call counters, random CDAN matrices) is kept in explicit ``dict``s.

Each function cites the reference file:line it follows (paths relative to the reference root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]
LayerSpec = List[Tuple[int, int, int]]          # [(in_ch, out_ch, kernel), ...] one omni-scale layer
NetSpec = List[LayerSpec]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------
# a1 — omni-scale layer spec                      OS_CNN/OS_CNN_Structure_build.py:3-42
# --------------------------------------------------------------------------------------
def prime_kernel_sizes(start: int, end: int) -> List[int]:
    """Kernel-size list.  ``1`` counts as prime because ``range(2, 1)`` is empty (:3-13)."""
    return [v for v in range(start, end + 1) if all(v % n != 0 for n in range(2, v))]


def omni_scale_spec(start: int, end: int, budgets: Sequence[int], in_channel: int = 1) -> NetSpec:
    """Layer list of (in, out, kernel) tuples; last layer is two branches k∈{start,start+1} (:20-42)."""
    primes = prime_kernel_sizes(start, end)
    spec: NetSpec = []
    cin = in_channel
    for budget in budgets:
        width = int(budget / (cin * sum(primes)))            # :16-18
        spec.append([(cin, width, p) for p in primes])
        cin = len(primes) * width
    w_last = len(primes) * int(budgets[0] / (in_channel * sum(primes)))   # :37-38
    spec.append([(cin, w_last, start), (cin, w_last, start + 1)])
    return spec


def respec_first_layer(spec: NetSpec, in_channel: int) -> NetSpec:
    """Classifier spec = same widths/kernels, first layer fed by ``in_channel`` (OS_CNN.py:142-152)."""
    return [[(in_channel, o, k) for (_, o, k) in spec[0]]] + [list(l) for l in spec[1:]]


def feature_width(spec: NetSpec) -> int:
    return sum(o for (_, o, _) in spec[-1])


def train_specs(length: int, in_channel: int) -> Tuple[NetSpec, NetSpec]:
    """(feature-extractor spec, classifier spec) as train_and_test.py:38-53 builds them."""
    budgets = [8 * 128 * in_channel, 5 * 128 * 256 + 2 * 256 * 128]
    rf = min(int(length / 4), 89)
    fe = omni_scale_spec(1, rf, budgets, in_channel)
    return fe, respec_first_layer(fe, feature_width(fe))


# --------------------------------------------------------------------------------------
# a2 — tap windows and masks                                      OS_CNN/OS_CNN.py:9-43
# --------------------------------------------------------------------------------------
def live_taps(kernel: int, kmax: int) -> Tuple[int, int]:
    """[lo, hi) of the Kmax window a ``kernel``-tap branch occupies (:9-12)."""
    right = math.ceil((kmax - 1) / 2) - math.ceil((kernel - 1) / 2)
    lo = kmax - kernel - right
    return lo, lo + kernel


def layer_mask(layer: LayerSpec) -> Tensor:
    """[ΣCout, Cin, Kmax] 0/1 mask, branches concatenated along out channels (:23-43)."""
    kmax = layer[-1][2]
    rows = []
    for cin, cout, k in layer:
        lo, hi = live_taps(k, kmax)
        m = torch.zeros(cout, cin, kmax)
        m[:, :, lo:hi] = 1.0
        rows.append(m)
    return torch.cat(rows, 0)


def same_pad(kmax: int) -> Tuple[int, int]:
    return int((kmax - 1) / 2), int(kmax / 2)                    # OS_CNN.py:59,158


# --------------------------------------------------------------------------------------
# a3 — one omni-scale layer                                     OS_CNN/OS_CNN.py:46-77
# --------------------------------------------------------------------------------------
def _batch_norm(y: Tensor, P: Params, pre: str, training: bool) -> Tensor:
    if training:
        P[pre + "num_batches_tracked"] += 1
    return F.batch_norm(y, P[pre + "running_mean"], P[pre + "running_var"], P[pre + "weight"],
                        P[pre + "bias"], training, BN_MOMENTUM, BN_EPS)


def omni_layer(x: Tensor, P: Params, pre: str, layer: LayerSpec, relu: bool, training: bool) -> Tensor:
    """masked Kmax conv (+bias) → BatchNorm1d → optional ReLU (:67-77).

    Q1: the reference re-masks ``weight.data`` every forward and convolves with the parameter
    itself, so ``dW`` is dense over Kmax.  Reproduced by masking ``.data`` in place.
    """
    kmax = layer[-1][2]
    w = P[pre + "conv1d.weight"]
    w.data.mul_(layer_mask(layer))
    y = F.conv1d(F.pad(x, same_pad(kmax)), w, P[pre + "conv1d.bias"])
    y = _batch_norm(y, P, pre + "bn.", training)
    return F.relu(y) if relu else y


# --------------------------------------------------------------------------------------
# a4/a5 — residual feature extractor OS_CNN_res               OS_CNN/OS_CNN.py:155-220
# --------------------------------------------------------------------------------------
SHARED_BLOCK_PREFIX = "net_1.net.net."


def feature_extractor(x: Tensor, P: Params, spec: NetSpec, training: bool) -> Tensor:
    h = x
    for i, layer in enumerate(spec):
        h = omni_layer(h, P, f"{SHARED_BLOCK_PREFIX}{i}.", layer, relu=(i != len(spec) - 1), training=training)
    s = F.conv1d(x, P["net_1.res.conv1d.weight"], P["net_1.res.conv1d.bias"])      # k=1, pad (0,0)
    s = _batch_norm(s, P, "net_1.res.bn.", training)
    return F.relu(s + h)                                                             # :176-180


def shared_block_params(P: Params, spec: NetSpec) -> List[Tensor]:
    """The 12 tensors ``return_last_layer().parameters()`` yields, in order (:219-220)."""
    out = []
    for i in range(len(spec)):
        for leaf in ("conv1d.weight", "conv1d.bias", "bn.weight", "bn.bias"):
            out.append(P[f"{SHARED_BLOCK_PREFIX}{i}.{leaf}"])
    return out


# --------------------------------------------------------------------------------------
# a6 — classifier OS_CNN                                       OS_CNN/OS_CNN.py:80-110
# --------------------------------------------------------------------------------------
def classifier(x: Tensor, P: Params, spec: NetSpec, training: bool) -> Tuple[Tensor, Tensor]:
    h = x
    for i, layer in enumerate(spec):
        h = omni_layer(h, P, f"net.{i}.", layer, relu=True, training=training)
    pooled = h.mean(dim=-1)                                       # AdaptiveAvgPool1d(1)+squeeze
    return F.linear(pooled, P["hidden.weight"], P["hidden.bias"]), pooled


# --------------------------------------------------------------------------------------
# a7-a11 — simplified WaveGlow                         Simplified_NF_WaveGlow.py:8-241
# --------------------------------------------------------------------------------------
WN_LAYERS = 8
WN_KERNEL = 3


def _wn_conv(x: Tensor, P: Params, pre: str, dilation: int = 1, padding: int = 0) -> Tensor:
    """Conv1d under old-style weight_norm: w = g·v/‖v‖ per output channel (:69-99)."""
    w = torch._weight_norm(P[pre + "weight_v"], P[pre + "weight_g"], 0)
    return F.conv1d(x, w, P[pre + "bias"], dilation=dilation, padding=padding)


def wn_forward(u0: Tensor, P: Params, pre: str) -> Tensor:
    """Gated dilated-conv stack conditioned on its own input (:101-123)."""
    n = P[pre + "start.weight_g"].shape[0]
    a = _wn_conv(u0, P, pre + "start.")
    out = torch.zeros_like(a)
    cond = _wn_conv(u0, P, pre + "cond_layer.")
    for i in range(WN_LAYERS):
        d = 2 ** i
        g = _wn_conv(a, P, f"{pre}in_layers.{i}.", dilation=d, padding=int((WN_KERNEL * d - d) / 2))
        g = g + cond[:, 2 * n * i: 2 * n * (i + 1)]
        acts = torch.tanh(g[:, :n]) * torch.sigmoid(g[:, n:])     # :44-54
        rs = _wn_conv(acts, P, f"{pre}res_skip_layers.{i}.")
        if i < WN_LAYERS - 1:
            a = a + rs[:, :n]
            out = out + rs[:, n:]
        else:
            out = out + rs
    return F.conv1d(out, P[pre + "end.weight"], P[pre + "end.bias"])


def waveglow_forward(x: Tensor, P: Params, n_flows: int) -> Tuple[Tensor, List[Tensor], List[Tensor]]:
    """(:149-181) returns (z, [log_s]·n_flows, [log_det_W]·n_flows)."""
    log_s_list, log_det_list = [], []
    B, C, L = x.shape
    h = C // 2
    for k in range(n_flows):
        W = P[f"convinv.{k}.conv.weight"]
        log_det_list.append(B * L * torch.logdet(W.squeeze()))    # :40
        x = F.conv1d(x, W)
        x0, x1 = x[:, :h], x[:, h:]
        o = wn_forward(x0, P, f"WN.{k}.")
        b, log_s = o[:, :h], o[:, h:]
        x1 = torch.exp(log_s) * x1 + b
        log_s_list.append(log_s)
        x = torch.cat([x0, x1], 1)
    return x, log_s_list, log_det_list


def waveglow_infer(z: Tensor, P: Params, n_flows: int, inv_cache: Dict[int, Tensor]) -> Tensor:
    """(:183-203).  Q2: ``W_inverse`` is computed on first use, detached, and cached forever."""
    h = z.shape[1] // 2
    x = z
    for k in reversed(range(n_flows)):
        x0, x1 = x[:, :h], x[:, h:]
        o = wn_forward(x0, P, f"WN.{k}.")
        b, s = o[:, :h], o[:, h:]
        x1 = (x1 - b) / torch.exp(s)
        x = torch.cat([x0, x1], 1)
        if k not in inv_cache:
            inv_cache[k] = P[f"convinv.{k}.conv.weight"].detach().squeeze().float().inverse()[..., None]
        x = F.conv1d(x, inv_cache[k])
    return x


def waveglow_loss(out: Tuple[Tensor, List[Tensor], List[Tensor]], sigma: float = 1.0) -> Tensor:
    z, log_s_list, log_det_list = out                             # :230-241
    log_s_total = sum(torch.sum(s) for s in log_s_list)
    log_det_total = sum(log_det_list)
    loss = torch.sum(z * z) / (2 * sigma * sigma) - log_s_total - log_det_total
    return loss / (z.size(0) * z.size(1) * z.size(2))


# --------------------------------------------------------------------------------------
# a12 — CPC InfoNCE                                    Comparison/SLARDA/train.py:41-76
# --------------------------------------------------------------------------------------
def cpc_nce(feat: Tensor, P: Params, timestep: int, t_samples: Optional[int] = None) -> Tensor:
    """Q6: the reference draws ``t_samples`` from the global CPU RNG; pass it to pin."""
    z = feat.transpose(1, 2)                                      # [B, L, C]
    B, _, C = z.shape
    if t_samples is None:
        t_samples = int(torch.randint(timestep // 2, size=(1,)))
    t = t_samples
    enc = z[:, t + 1: t + 1 + timestep, :].transpose(0, 1)        # [T, B, C]
    hid = P["gru.weight_hh_l0"].shape[1]
    h0 = torch.zeros(1, B, hid)
    out, _ = torch._VF.gru(z[:, : t + 1, :], h0,
                           [P["gru.weight_ih_l0"], P["gru.weight_hh_l0"], P["gru.bias_ih_l0"], P["gru.bias_hh_l0"]],
                           True, 1, 0.0, False, False, True)
    c_t = out[:, t, :]
    nce = 0.0
    for i in range(timestep):
        pred = F.linear(c_t, P[f"Wk.{i}.weight"], P[f"Wk.{i}.bias"])
        total = enc[i] @ pred.t()
        nce = nce + torch.sum(torch.diag(F.log_softmax(total, dim=-1)))
    return nce / (-1.0 * B * timestep)


# --------------------------------------------------------------------------------------
# a13-a15 — CDAN                                    C_DAN.py:11-82, widgets.py:8-13,95-131
# --------------------------------------------------------------------------------------
def grl_coeff(iter_num: float, high=1.0, low=0.0, alpha=100.0, max_iter=20.0) -> float:
    return float(2.0 * (high - low) / (1.0 + np.exp(-alpha * iter_num / max_iter)) - (high - low) + low)


def _grl(x: Tensor, coeff: float) -> Tensor:
    y = x * 1.0
    if y.requires_grad:
        y.register_hook(lambda g: -coeff * g.clone())
    return y


def _bump(counter: Dict[str, float], training: bool) -> float:
    """GRL call counter, shared logic of ad_net and fd_s (widgets.py:34-38,115-119)."""
    if training:
        counter["iter_num"] += 1
    if counter["iter_num"] >= counter["max_iter"]:
        counter["iter_num"] = counter["max_iter"]
    c = grl_coeff(counter["iter_num"], alpha=counter["alpha"], max_iter=counter["max_iter"])
    counter["coeff"] = c
    return c


def new_grl_counter() -> Dict[str, float]:
    return {"iter_num": -1, "alpha": 100.0, "max_iter": 20.0, "coeff": 0.001}


def ad_net_forward(x: Tensor, P: Params, counter: Dict[str, float], training: bool, dropout_p: float = 0.2) -> Tensor:
    c = _bump(counter, training)
    h = _grl(x, c)
    h = F.dropout(F.relu(F.linear(h, P["ad_layer1.weight"], P["ad_layer1.bias"])), dropout_p, training)
    h = F.dropout(F.relu(F.linear(h, P["ad_layer2.weight"], P["ad_layer2.bias"])), dropout_p, training)
    return F.linear(h, P["ad_layer3.weight"], P["ad_layer3.bias"])


def random_layer(xs: Sequence[Tensor], mats: Sequence[Tensor]) -> Tensor:
    outs = [x @ m for x, m in zip(xs, mats)]                      # C_DAN.py:20-25
    r = outs[0] / math.pow(float(mats[0].shape[1]), 1.0 / len(outs))
    for o in outs[1:]:
        r = r * o
    return r


def entropy(p: Tensor) -> Tensor:
    return torch.sum(-p * torch.log(p + 1e-5), dim=1)             # C_DAN.py:32-37


def cdan_loss(feat_t: Tensor, feat_g: Tensor, logit_t: Tensor, logit_g: Tensor, P_ad: Params,
              counter: Dict[str, float], mats: Sequence[Tensor], training: bool, dropout_p: float = 0.2) -> Tensor:
    """C_DAN.py:49-82 with the random multilinear map.  Q4: ``[B]*[B,1]`` broadcasts to ``[B,B]``."""
    xt, xg = torch.flatten(feat_t, 1), torch.flatten(feat_g, 1)
    pt, pg = F.softmax(logit_t, dim=1), F.softmax(logit_g, dim=1)
    out_t = ad_net_forward(random_layer([xt, pt], mats), P_ad, counter, training, dropout_p)
    out_g = ad_net_forward(random_layer([xg, pg], mats), P_ad, counter, training, dropout_p)
    coeff = counter["coeff"]
    ent_t, ent_g = entropy(pt), entropy(pg)
    if ent_t.requires_grad:
        ent_t.register_hook(lambda g: -coeff * g.clone())
    if ent_g.requires_grad:
        ent_g.register_hook(lambda g: -coeff * g.clone())
    w_t = 1.0 + torch.exp(-ent_t)
    w_g = 1.0 + torch.exp(-ent_g)
    w_t = w_t / torch.sum(w_t).detach().item()
    w_g = w_g / torch.sum(w_g).detach().item()
    return torch.sum(w_t * out_t) - torch.sum(w_g * out_g)


# --------------------------------------------------------------------------------------
# a16-a18 — small heads                                              widgets.py:15-167
# --------------------------------------------------------------------------------------
def new_noise_state(channels: int, length: int) -> Dict[str, object]:
    return {"target_avg": torch.zeros(channels, length), "source_avg": torch.zeros(channels, length),
            "time": 0, "n_t": 0, "n_s": 0}


def noise_transfer(z_t: Tensor, z_s: Tensor, P: Params, st: Dict[str, object]) -> Tensor:
    """Q5: the running "averages" are not averages; state is detached after every call (:150-167)."""
    st["time"] += 1
    bt, bs = z_t.size(0), z_s.size(0)
    if st["time"] == 1:
        st["target_avg"] = st["target_avg"] + torch.mean(z_t, dim=0)
        st["source_avg"] = st["source_avg"] + torch.mean(z_s, dim=0)
    else:
        st["target_avg"] = st["target_avg"] + (bt / st["n_t"]) * torch.mean(z_t, dim=0)
        st["source_avg"] = st["source_avg"] + (bs / st["n_s"]) * torch.mean(z_s, dim=0)
    st["n_t"] += bt
    st["n_s"] += bs
    dist = st["target_avg"] - st["source_avg"]
    learned = F.selu(F.conv1d(dist, P["apply_learnable_weight.weight"], P["apply_learnable_weight.bias"]))
    st["source_avg"] = st["source_avg"].detach()
    st["target_avg"] = st["target_avg"].detach()
    return learned + z_s


def dimension_unification(x: Tensor, P: Params) -> Tensor:
    h = F.relu(F.linear(x, P["length_unification.weight"], P["length_unification.bias"]))    # :73-78
    return F.relu(F.conv1d(h, P["channel_unification.weight"], P["channel_unification.bias"]))


def prob_transfer(pooled: Tensor, P: Params) -> Tensor:
    """LSTM over the input repeated twice, returns h_n (:51-55)."""
    x = torch.stack([pooled, pooled], dim=1)
    B, H = pooled.shape
    zeros = torch.zeros(1, B, H)
    _, h_n, _ = torch._VF.lstm(x, (zeros, zeros),
                               [P["model.weight_ih_l0"], P["model.weight_hh_l0"], P["model.bias_ih_l0"], P["model.bias_hh_l0"]],
                               True, 1, 0.0, False, False, True)
    return h_n.squeeze(0)


def feature_discriminator(x: Tensor, P: Params, counter: Dict[str, float], training: bool) -> Tensor:
    c = _bump(counter, training)                                  # :32-42
    h = _grl(x, c)
    for i in (0, 2, 4):
        h = F.leaky_relu(F.linear(h, P[f"model.{i}.weight"], P[f"model.{i}.bias"]), 0.2)
    return F.linear(h, P["model.6.weight"], P["model.6.bias"])


def wgan_loss(v_t: Tensor, v_s2t2s: Tensor, v_s: Tensor) -> Tensor:
    return -torch.mean(v_t) - torch.mean(v_s2t2s) + torch.mean(v_s)   # :59-61


# --------------------------------------------------------------------------------------
# a19 — the train steps                                     train_and_test.py:141-766
# --------------------------------------------------------------------------------------
def loss_coefficients(epoch: int) -> Tuple[float, float, float, float]:
    """(a·cdan, b·fd_s, c·sl_t, d·sl_s) by epoch (train_and_test.py:665-672)."""
    if epoch < 12:
        return 3, 3, 2, 2
    if epoch < 24:
        return 2, 3, 1.8, 1.5
    if epoch < 50:
        return 1.5, 2, 1.8, 1.8
    return 1.5, 1.5, 2.5, 2.5


def _leaves(P: Params) -> List[Tensor]:
    return [t for t in P.values() if t.requires_grad]


class ClassifierStep:
    """S1: FE → CLF → CE → backward → RMSprop×2 (train_and_test.py:148-171 without the CPC term)."""

    def __init__(self, fe: Params, clf: Params, fe_spec: NetSpec, clf_spec: NetSpec):
        self.fe, self.clf, self.fe_spec, self.clf_spec = fe, clf, fe_spec, clf_spec
        self.opt_fe = torch.optim.RMSprop(_leaves(fe), lr=0.001)
        self.opt_clf = torch.optim.RMSprop(_leaves(clf), lr=0.003)

    def step(self, x: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
        feat = feature_extractor(x, self.fe, self.fe_spec, True)
        logits, _ = classifier(feat, self.clf, self.clf_spec, True)
        loss = F.cross_entropy(logits, y)
        loss.backward()
        self.opt_fe.step(); self.opt_clf.step()
        self.opt_fe.zero_grad(); self.opt_clf.zero_grad()
        return loss.detach(), logits.detach()


class JointStep:
    """S2: one batch of the joint phase incl. GradNorm (train_and_test.py:539-766), literally.

    ``mods`` maps module name → Params for: fe_t, clf_t, fe_s, dimunif, clf_s, probtransfer, nf,
    noise, ad_net, fd_s, cpc.  ``mats`` are the two fixed CDAN random matrices.
    """

    LRS = {"fe_t": 0.001, "clf_t": 0.003, "fe_s": 0.001, "dimunif": 0.001, "clf_s": 0.003,
           "probtransfer": 0.001, "nf": 0.001, "noise": 0.005, "ad_net": 0.001, "fd_s": 0.001}

    def __init__(self, mods: Dict[str, Params], mats: Sequence[Tensor], fe_t_spec: NetSpec, clf_spec: NetSpec,
                 fe_s_spec: NetSpec, n_flows: int, cpc_timestep: int, dropout_p: float = 0.2):
        self.m, self.mats = mods, list(mats)
        self.fe_t_spec, self.clf_spec, self.fe_s_spec = fe_t_spec, clf_spec, fe_s_spec
        self.n_flows, self.T, self.dropout_p = n_flows, cpc_timestep, dropout_p
        self.opts = {k: torch.optim.RMSprop(_leaves(mods[k]), lr=lr) for k, lr in self.LRS.items()}
        self.opt_cpc = torch.optim.Adam(_leaves(mods["cpc"]), lr=0.002)
        self.w_t = torch.tensor([2.0, 5.0], requires_grad=True)                    # :501-505
        self.w_s = torch.tensor([2.0, 2.0, 4.0], requires_grad=True)
        self.opt_w_t = torch.optim.Adam([self.w_t], lr=0.0002)
        self.opt_w_s = torch.optim.Adam([self.w_s], lr=0.001)
        self.init_t = self.init_s = None
        self.alpha = 3
        C = feature_width(fe_t_spec)
        L = mods["dimunif"]["length_unification.weight"].shape[0]
        self.noise_state = new_noise_state(C, L)
        self.inv_cache: Dict[int, Tensor] = {}
        self.ad_counter, self.fd_counter = new_grl_counter(), new_grl_counter()

    def forward_losses(self, x_t, y_t, x_s, y_s, t_samples: Tuple[Optional[int], Optional[int]] = (None, None)):
        m = self.m
        feat_t = feature_extractor(x_t, m["fe_t"], self.fe_t_spec, True)
        sl_t = cpc_nce(feat_t, m["cpc"], self.T, t_samples[0])
        feat_s = dimension_unification(feature_extractor(x_s, m["fe_s"], self.fe_s_spec, True), m["dimunif"])
        sl_s = cpc_nce(feat_s, m["cpc"], self.T, t_samples[1])
        nf_t_out = waveglow_forward(feat_t, m["nf"], self.n_flows)
        nf_s_out = waveglow_forward(feat_s, m["nf"], self.n_flows)
        nf_t, nf_s = waveglow_loss(nf_t_out), waveglow_loss(nf_s_out)
        z_s2t = noise_transfer(nf_t_out[0], nf_s_out[0], m["noise"], self.noise_state)
        feat_s2t = waveglow_infer(z_s2t, m["nf"], self.n_flows, self.inv_cache)
        logit_t, pooled_t = classifier(feat_t, m["clf_t"], self.clf_spec, True)
        logit_s2t, pooled_s2t = classifier(feat_s2t, m["clf_t"], self.clf_spec, False)   # :584-586 eval mode
        logit_s, pooled_s = classifier(feat_s, m["clf_s"], self.clf_spec, True)
        ce_t, ce_s = F.cross_entropy(logit_t, y_t), F.cross_entropy(logit_s, y_s)
        cdan = cdan_loss(feat_t, feat_s2t, logit_t, logit_s2t, m["ad_net"], self.ad_counter, self.mats, True,
                         self.dropout_p)
        tr_t = prob_transfer(pooled_t, m["probtransfer"])
        tr_s2t = prob_transfer(pooled_s2t, m["probtransfer"])
        logit_s2t2s = F.linear(tr_s2t, m["clf_s"]["hidden.weight"], m["clf_s"]["hidden.bias"])
        ce_s2t2s = F.cross_entropy(logit_s2t2s, y_s)
        fd = wgan_loss(feature_discriminator(tr_t, m["fd_s"], self.fd_counter, True),
                       feature_discriminator(tr_s2t, m["fd_s"], self.fd_counter, True),
                       feature_discriminator(pooled_s, m["fd_s"], self.fd_counter, True))
        losses = {"nf_t": nf_t, "nf_s": nf_s, "ce_t": ce_t, "sl_t": sl_t, "ce_s": ce_s, "sl_s": sl_s,
                  "cdan": cdan, "ce_s2t2s": ce_s2t2s, "fd_s": fd}
        aux = {"logit_t": logit_t, "logit_s": logit_s, "logit_s2t": logit_s2t, "feat_t": feat_t, "feat_s2t": feat_s2t}
        return losses, aux

    # the pre-training phases of train() — sub-graphs of the joint step on the same modules and optimisers
    PHASES = {                                                                       # phase -> optimisers stepped
        "target_pretrain": ("fe_t", "clf_t", "cpc"),                                 # train_and_test.py:143-171
        "source_pretrain": ("fe_s", "dimunif", "clf_s"),                             # :183-209
        "ssl_with_ce": ("fe_t", "clf_t", "cpc", "fe_s", "dimunif", "clf_s"),         # :232-275 (every 50th epoch)
        "ssl": ("fe_t", "cpc", "fe_s", "dimunif"),                                   # :296-348
        "nf_with_ce": ("fe_t", "clf_t", "fe_s", "dimunif", "clf_s", "nf", "cpc"),    # :388-431 (every 75th epoch)
        "nf": ("fe_t", "fe_s", "dimunif", "nf"),                                     # :457-494 (features detached)
    }

    def phase_losses(self, phase: str, x_t, y_t, x_s, y_s, t_samples=(None, None)):
        """(total, losses) of one batch of a pre-training phase.  Note "ssl": both classifiers run in train mode
        (their BatchNorm running statistics move) although their losses are not in the total and they are not
        stepped (:308-324); "nf": the features are detached, so only the flow receives gradients (:467-471)."""
        m, L = self.m, {}
        if phase == "target_pretrain":
            feat_t = feature_extractor(x_t, m["fe_t"], self.fe_t_spec, True)
            L["sl_t"] = cpc_nce(feat_t, m["cpc"], self.T, t_samples[0])
            L["ce_t"] = F.cross_entropy(classifier(feat_t, m["clf_t"], self.clf_spec, True)[0], y_t)
            return L["ce_t"] + L["sl_t"], L
        if phase == "source_pretrain":
            feat_s = dimension_unification(feature_extractor(x_s, m["fe_s"], self.fe_s_spec, True), m["dimunif"])
            L["ce_s"] = F.cross_entropy(classifier(feat_s, m["clf_s"], self.clf_spec, True)[0], y_s)
            return L["ce_s"], L
        feat_t = feature_extractor(x_t, m["fe_t"], self.fe_t_spec, True)
        feat_s = dimension_unification(feature_extractor(x_s, m["fe_s"], self.fe_s_spec, True), m["dimunif"])
        if phase == "nf":
            feat_t, feat_s = feat_t.detach(), feat_s.detach()
        else:
            L["sl_t"] = cpc_nce(feat_t, m["cpc"], self.T, t_samples[0])
            L["ce_t"] = F.cross_entropy(classifier(feat_t, m["clf_t"], self.clf_spec, True)[0], y_t)
            L["sl_s"] = cpc_nce(feat_s, m["cpc"], self.T, t_samples[1])
            L["ce_s"] = F.cross_entropy(classifier(feat_s, m["clf_s"], self.clf_spec, True)[0], y_s)
        if phase == "ssl_with_ce":
            return L["sl_t"] + L["sl_s"] + 0.8 * L["ce_t"] + 1.2 * L["ce_s"], L
        if phase == "ssl":
            return L["sl_t"] + L["sl_s"], L
        L["nf_t"] = waveglow_loss(waveglow_forward(feat_t, m["nf"], self.n_flows))
        L["nf_s"] = waveglow_loss(waveglow_forward(feat_s, m["nf"], self.n_flows))
        if phase == "nf_with_ce":
            return L["nf_t"] + L["nf_s"] + 5 * L["ce_t"] + 5 * L["ce_s"] + 3 * L["sl_t"] + 3 * L["sl_s"], L
        if phase == "nf":
            return L["nf_t"] + L["nf_s"], L
        raise ValueError(f"unknown phase {phase!r}")

    def phase_step(self, phase: str, x_t, y_t, x_s, y_s, t_samples=(None, None)):
        total, L = self.phase_losses(phase, x_t, y_t, x_s, y_s, t_samples)
        total.backward()
        for k in self.PHASES[phase]:
            (self.opt_cpc if k == "cpc" else self.opts[k]).step()
        report = {k: v.detach().clone() for k, v in L.items()}
        report["total"] = total.detach().clone()
        return report

    def zero_grad(self):
        for o in self.opts.values():
            o.zero_grad()
        self.opt_cpc.zero_grad()

    def step(self, x_t, y_t, x_s, y_s, epoch: int = 0, t_samples=(None, None)):
        L, aux = self.forward_losses(x_t, y_t, x_s, y_s, t_samples)
        report = {k: v.detach().clone() for k, v in L.items()}
        lt = torch.stack([L["nf_t"], L["ce_t"]])
        ls = torch.stack([L["nf_s"], L["ce_s"], L["ce_s2t2s"]])
        if self.init_t is None:                                                     # :658-664
            self.init_t = 1 / (1 + np.exp(-lt.data.numpy()))
            self.init_s = 1 / (1 + np.exp(-ls.data.numpy()))
        a, b, c, d = loss_coefficients(epoch)
        cdan, fd = L["cdan"], L["fd_s"]
        total = torch.sum(self.w_t * lt) + torch.sum(self.w_s * ls) + a * cdan + b * fd + c * L["sl_t"] + d * L["sl_s"]
        for o in self.opts.values():
            o.zero_grad()
        self.opt_cpc.zero_grad(); self.opt_w_s.zero_grad(); self.opt_w_t.zero_grad()
        total.backward(retain_graph=True)                                           # :678
        self.opt_w_s.zero_grad(); self.opt_w_t.zero_grad()
        sh_t = shared_block_params(self.m["fe_t"], self.fe_t_spec)
        sh_s = shared_block_params(self.m["fe_s"], self.fe_s_spec)
        norms_t = [torch.cat([torch.norm(self.w_t[i] * g).unsqueeze(0)
                              for g in torch.autograd.grad(lt[i], sh_t, retain_graph=True)]).sum() for i in range(2)]
        norms_s = [torch.cat([torch.norm(self.w_s[i] * g).unsqueeze(0)
                              for g in torch.autograd.grad(ls[i], sh_s, retain_graph=True)]).sum() for i in range(3)]
        nt, ns = torch.stack(norms_t), torch.stack(norms_s)
        ratio_t = (1 / (1 + np.exp(-lt.data.numpy()))) / self.init_t                # :694-700
        ratio_s = (1 / (1 + np.exp(-ls.data.numpy()))) / self.init_s
        inv_t, inv_s = ratio_t / np.mean(ratio_t), ratio_s / np.mean(ratio_s)
        const_t = torch.tensor(np.mean(nt.data.numpy()) * (inv_t ** self.alpha), requires_grad=False)
        const_s = torch.tensor(np.mean(ns.data.numpy()) * (inv_s ** self.alpha), requires_grad=False)
        g_w_t = torch.autograd.grad(torch.sum(torch.abs(nt - const_t)), self.w_t)[0]
        g_w_s = torch.autograd.grad(torch.sum(torch.abs(ns - const_s)), self.w_s)[0]
        saved_t, saved_s = self.w_t.data.numpy().copy(), self.w_s.data.numpy().copy()
        # Q3 (:734-741): zero the .data of losses/weights, then backward through the SAME graph again.
        total.data = total.data * 0.0
        self.w_t.data = self.w_t.data * 0.0
        self.w_s.data = self.w_s.data * 0.0
        lt.data = lt.data * 0.0
        ls.data = ls.data * 0.0
        cdan.data = cdan.data * 0.0
        fd.data = fd.data * 0.0
        total.backward()
        self.w_t.data = torch.tensor(saved_t)
        self.w_s.data = torch.tensor(saved_s)
        self.w_t.grad, self.w_s.grad = g_w_t, g_w_s
        self.opt_w_t.step(); self.opt_w_s.step()
        for o in self.opts.values():
            o.step()
        self.opt_cpc.step()
        self.w_t.data.clamp_(min=0.0)
        self.w_t.data = self.w_t.data * (7 / torch.sum(self.w_t.data, dim=0))
        self.w_s.data.clamp_(min=0.0)
        self.w_s.data = self.w_s.data * (8 / torch.sum(self.w_s.data, dim=0))
        for p in _leaves(self.m["ad_net"]):
            p.data.clamp_(-0.0005, 0.0005)
        for p in _leaves(self.m["fd_s"]):
            p.data.clamp_(-0.01, 0.01)
        report["w_t"], report["w_s"] = self.w_t.data.clone(), self.w_s.data.clone()
        report["norms_t"], report["norms_s"] = nt.detach().clone(), ns.detach().clone()
        report.update({k: v.detach().clone() for k, v in aux.items()})
        return report


# --------------------------------------------------------------------------------------
# helpers for fixtures / baselines: fresh parameter dicts with the reference's shapes + init laws
# --------------------------------------------------------------------------------------
def to_params(state: Dict[str, np.ndarray]) -> Params:
    """npz/state_dict → leaf tensors (floating tensors require grad, except BN running stats)."""
    P: Params = {}
    for k, v in state.items():
        t = torch.as_tensor(np.asarray(v)).clone()
        if k.endswith("num_batches_tracked"):
            t = t.long()
        if t.is_floating_point() and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        P[k] = t
    return P


def _conv_init(cout: int, cin: int, k: int, gen: torch.Generator) -> Tuple[Tensor, Tensor]:
    bound = 1.0 / math.sqrt(cin * k)                              # kaiming_uniform(a=√5) ⇒ U(±1/√fan_in)
    w = (torch.rand(cout, cin, k, generator=gen) * 2 - 1) * bound
    b = (torch.rand(cout, generator=gen) * 2 - 1) * bound
    return w, b


def _bn_init(P: Dict[str, Tensor], pre: str, c: int) -> None:
    P[pre + "weight"], P[pre + "bias"] = torch.ones(c), torch.zeros(c)
    P[pre + "running_mean"], P[pre + "running_var"] = torch.zeros(c), torch.ones(c)
    P[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _omni_layer_init(P: Dict[str, Tensor], pre: str, layer: LayerSpec, gen: torch.Generator) -> None:
    kmax = layer[-1][2]
    ws, bs = [], []
    for cin, cout, k in layer:                                    # per-branch fan-in (Q9)
        w, b = _conv_init(cout, cin, k, gen)
        lo, hi = live_taps(k, kmax)
        big = torch.zeros(cout, cin, kmax)
        big[:, :, lo:hi] = w
        ws.append(big); bs.append(b)
    P[pre + "conv1d.weight"], P[pre + "conv1d.bias"] = torch.cat(ws, 0), torch.cat(bs, 0)
    _bn_init(P, pre + "bn.", P[pre + "conv1d.bias"].numel())


def _finish(P: Dict[str, Tensor]) -> Params:
    return to_params({k: v.numpy() for k, v in P.items()})


def init_feature_extractor(spec: NetSpec, gen: torch.Generator) -> Params:
    P: Dict[str, Tensor] = {}
    for i, layer in enumerate(spec):
        _omni_layer_init(P, f"{SHARED_BLOCK_PREFIX}{i}.", layer, gen)
    C = feature_width(spec)
    P["net_1.res.conv1d.weight"], P["net_1.res.conv1d.bias"] = _conv_init(C, spec[0][0][0], 1, gen)
    _bn_init(P, "net_1.res.bn.", C)
    return _finish(P)


def init_classifier(spec: NetSpec, n_class: int, gen: torch.Generator) -> Params:
    P: Dict[str, Tensor] = {}
    for i, layer in enumerate(spec):
        _omni_layer_init(P, f"net.{i}.", layer, gen)
    C = feature_width(spec)
    w, b = _conv_init(n_class, C, 1, gen)
    P["hidden.weight"], P["hidden.bias"] = w[:, :, 0], b
    return _finish(P)


def _wn_conv_init(P, pre, cout, cin, k, gen):
    w, b = _conv_init(cout, cin, k, gen)
    P[pre + "weight_v"], P[pre + "bias"] = w, b
    P[pre + "weight_g"] = w.flatten(1).norm(dim=1).view(cout, 1, 1)


def init_waveglow(n_flows: int, n_group: int, n_channels: int, gen: torch.Generator, zero_end: bool = True) -> Params:
    P: Dict[str, Tensor] = {}
    h = n_group // 2
    for k in range(n_flows):
        q = torch.linalg.qr(torch.randn(n_group, n_group, generator=gen))[0]
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        P[f"convinv.{k}.conv.weight"] = q.contiguous().view(n_group, n_group, 1)
        pre = f"WN.{k}."
        _wn_conv_init(P, pre + "start.", n_channels, h, 1, gen)
        _wn_conv_init(P, pre + "cond_layer.", 2 * n_channels * WN_LAYERS, h, 1, gen)
        for i in range(WN_LAYERS):
            _wn_conv_init(P, f"{pre}in_layers.{i}.", 2 * n_channels, n_channels, WN_KERNEL, gen)
            rs = 2 * n_channels if i < WN_LAYERS - 1 else n_channels
            _wn_conv_init(P, f"{pre}res_skip_layers.{i}.", rs, n_channels, 1, gen)
        w, b = _conv_init(2 * h, n_channels, 1, gen)
        if zero_end:                                              # reference zero-inits ``end`` (:75-77)
            w, b = torch.zeros_like(w), torch.zeros_like(b)
        else:                                                     # non-degenerate flows for benchmarks/tests
            w, b = 0.05 * w, 0.05 * b
        P[pre + "end.weight"], P[pre + "end.bias"] = w, b
    return _finish(P)


def _linear_init(P, pre, cout, cin, gen, xavier=False):
    if xavier:                                                    # widgets.py:82-92
        P[pre + "weight"] = torch.randn(cout, cin, generator=gen) * math.sqrt(2.0 / (cin + cout))
        P[pre + "bias"] = torch.zeros(cout)
    else:
        w, b = _conv_init(cout, cin, 1, gen)
        P[pre + "weight"], P[pre + "bias"] = w[:, :, 0], b


def init_cpc(channels: int, hidden: int, timestep: int, gen: torch.Generator) -> Params:
    P: Dict[str, Tensor] = {}
    bound = 1.0 / math.sqrt(hidden)
    for name, shape in (("weight_ih_l0", (3 * hidden, channels)), ("weight_hh_l0", (3 * hidden, hidden)),
                        ("bias_ih_l0", (3 * hidden,)), ("bias_hh_l0", (3 * hidden,))):
        P["gru." + name] = (torch.rand(*shape, generator=gen) * 2 - 1) * bound
    for i in range(timestep):
        _linear_init(P, f"Wk.{i}.", channels, hidden, gen)
    return _finish(P)


def init_small_heads(C: int, C_s: int, L_t: int, L_s: int, gen: torch.Generator) -> Dict[str, Params]:
    out: Dict[str, Dict[str, Tensor]] = {k: {} for k in ("dimunif", "probtransfer", "noise", "ad_net", "fd_s")}
    _linear_init(out["dimunif"], "length_unification.", L_t, L_s, gen)
    w, b = _conv_init(C, C_s, 1, gen)
    out["dimunif"]["channel_unification.weight"], out["dimunif"]["channel_unification.bias"] = w, b
    bound = 1.0 / math.sqrt(C)
    for name, shape in (("weight_ih_l0", (4 * C, C)), ("weight_hh_l0", (4 * C, C)),
                        ("bias_ih_l0", (4 * C,)), ("bias_hh_l0", (4 * C,))):
        out["probtransfer"]["model." + name] = (torch.rand(*shape, generator=gen) * 2 - 1) * bound
    w, b = _conv_init(C, C, 1, gen)
    out["noise"]["apply_learnable_weight.weight"], out["noise"]["apply_learnable_weight.bias"] = w, b
    _linear_init(out["ad_net"], "ad_layer1.", 1024, 1024, gen, xavier=True)
    _linear_init(out["ad_net"], "ad_layer2.", 1024, 1024, gen, xavier=True)
    _linear_init(out["ad_net"], "ad_layer3.", 1, 1024, gen, xavier=True)
    for i, (co, ci) in zip((0, 2, 4, 6), ((800, C), (400, 800), (50, 400), (1, 50))):
        _linear_init(out["fd_s"], f"model.{i}.", co, ci, gen)
    return {k: _finish(v) for k, v in out.items()}


def build_joint_step(L_t: int, C_in_t: int, L_s: int, C_in_s: int, n_class_t: int, n_class_s: int,
                     seed: int = 1234, nf_channels: int = 120, dropout_p: float = 0.2,
                     zero_end: bool = False) -> JointStep:
    """Fresh random-init joint step with the shapes train_and_test.py:26-134 derives."""
    gen = torch.Generator().manual_seed(seed)
    fe_t_spec, clf_spec = train_specs(L_t, C_in_t)
    fe_s_spec, _ = train_specs(L_s, C_in_s)
    C, C_s = feature_width(fe_t_spec), feature_width(fe_s_spec)
    mods = {"fe_t": init_feature_extractor(fe_t_spec, gen), "clf_t": init_classifier(clf_spec, n_class_t, gen),
            "fe_s": init_feature_extractor(fe_s_spec, gen), "clf_s": init_classifier(clf_spec, n_class_s, gen),
            "nf": init_waveglow(3, C, nf_channels, gen, zero_end=zero_end), "cpc": init_cpc(C, 64, L_t // 2, gen)}
    mods.update(init_small_heads(C, C_s, L_t, L_s, gen))
    mats = [torch.randn(C * L_t, 1024, generator=gen), torch.randn(n_class_t, 1024, generator=gen)]
    return JointStep(mods, mats, fe_t_spec, clf_spec, fe_s_spec, 3, L_t // 2, dropout_p)


# --------------------------------------------------------------------------------------
# multi-source voting                                   multi_source_voting.py:281-424
# --------------------------------------------------------------------------------------
def voting_precision_weights(train_logits: np.ndarray, train_labels: np.ndarray) -> np.ndarray:
    """Per model k and class c: of the train samples model k predicts as c, the fraction that is c (0 if it never
    predicts c) (:292-357); then every model's vector is divided by the mean over models, 0/0 -> nan -> 0 (:358-367)."""
    K, N, C = train_logits.shape
    w = np.zeros((K, C))
    for k in range(K):
        pred = np.argmax(train_logits[k], axis=1)
        for c in range(C):
            n_c = int(np.sum(pred == c))
            w[k, c] = (np.sum((pred == c) & (pred == train_labels)) / n_c) if n_c else 0.0
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.nan_to_num(w / w.mean(axis=0))
    return w


def voting_scores(test_logits: np.ndarray, weights: np.ndarray) -> np.ndarray:
    """Σ_k softmax(logits_k) · (1 + 120·exp(−H(softmax))) · 9^{w_k}   (:406-422); H = natural-log Shannon entropy."""
    K, M, C = test_logits.shape
    total = np.zeros((M, C), dtype=np.float32)
    for k in range(K):
        rows = test_logits[k].astype(np.float32).copy()
        for i in range(M):
            p = np.exp(rows[i]) / np.sum(np.exp(rows[i]))
            H = -np.sum(np.where(p > 0, p * np.log(p), 0.0))
            rows[i] = p * (1 + 120 * np.exp(-H)) * np.power(9, weights[k])
        total += rows
    return total


def multi_source_vote(train_logits, train_labels, test_logits, test_labels=None):
    w = voting_precision_weights(np.asarray(train_logits), np.asarray(train_labels))
    scores = voting_scores(np.asarray(test_logits), w)
    pred = np.argmax(scores, axis=1)
    acc = None if test_labels is None else float(np.mean(pred == np.asarray(test_labels)))
    return w, scores, pred, acc
