"""The WHOLE joint step (train_and_test.py:539-766) at the sizes BASELINE.json names, forward AND backward:

* configs[1] geometry (univariate, L=512): nine losses, logits, GradNorm norms/weights and every accumulated gradient
  (the reference's double backward, quirk Q3) against the CPU oracle at B=4;
* the same at configs[4]'s length (L=1024), B=2;
* configs[2]: four source domains -> four independent pipelines trained one step each -> K=4 eval forward -> vote
  (main.py:7-11, multi_source_voting.py:265-277,281-424), every pipeline's step against the oracle and the vote
  against the restated voting block on oracle logits;
* at the metric's full batch (B=256, L=512): capture() + replay() == step() from the same snapshot.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import feature_level_style_transfer_for_tsc_amd as fst
from oracle import restatement as R
from test_gpu_modules import check_grads, close

DEV = "cuda"
LOSSES = ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s", "cdan", "ce_s2t2s", "fd_s")


def _pair(gen, B, C_in, L, ncls):
    x = torch.randn(B, C_in, L, generator=gen)
    x = (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)
    return x, torch.randint(ncls, (B,), generator=gen)


def _trainer_from(js, L_t, L_s, ncls):
    cfg = fst.JointConfig(L_t=L_t, C_in_t=1, L_s=L_s, C_in_s=1, n_class_t=ncls, n_class_s=ncls, dropout_p=0.0)
    tr = fst.JointTrainer(cfg, DEV)
    tr.load_params({k: {n: t.detach() for n, t in v.items()} for k, v in js.m.items()}, js.mats)
    return tr


def _step_both(js, tr, batch, ts):
    """One step of the oracle and of the HIP trainer from identical state; returns (oracle report, oracle grads,
    trainer report, trainer grads) with the gradients taken right before the optimisers consume them."""
    (x_t, y_t), (x_s, y_s) = batch
    rep_o = js.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=ts)
    want = {name: {n: p.grad.detach().numpy().copy() for n, p in P.items() if p.requires_grad and p.grad is not None}
            for name, P in js.m.items()}
    grads = {}

    def grab():
        for name in tr.MODULES:
            grads[name] = {n: p.grad.detach().clone() for n, p in tr.m[name].named_parameters() if p.grad is not None}
    tr.on_grads_ready = grab
    rep = tr.step(x_t.to(DEV), y_t.to(DEV), x_s.to(DEV), y_s.to(DEV), epoch=0, t_samples=ts)
    tr.on_grads_ready = None
    return rep_o, want, rep, grads


# Gradient tolerances (of each module's gradient scale).
# f32 MFMA mode: measured 1e-4 and better when no unit flips.
# Split-bf16 mode (pre-activations differ from the oracle's by ~5e-6 of their scale instead of 1e-7):
#  * 3e-3 for modules that are not upstream of a ReLU head (measured <= 4e-4; conv biases in front of a train-mode
#    BatchNorm have a mathematically zero gradient: Σ dy with Σ dy = 0, pure rounding residue, 2e-3 at B=3);
#  * 5e-2 for everything upstream of the adversarial MLPs (ad_net: 2 x 1024 ReLU units per sample, fd_s): with 3-4 samples
#    per batch one unit within rounding distance of zero takes the other branch in about one run in five (the K-split
#    fp32 atomics of the random-layer GEMM move the last bits between runs), which changes the cotangent entering the
#    flow / extractors by ~1e-2 of its scale — tests/diag_grad_vs_oracle.py shows the pattern: ad_layer1 and every module
#    upstream of it at 1e-2, ad_layer2/3, clf_s, probtransfer, fd_s, cpc at 1e-5, the f32 mode at 1e-6 throughout.
#    (DESIGN.md "gradients of a ReLU network are only piecewise comparable".)  The kernels themselves are held to fp64
#    references in tests/test_gpu_kernels.py and the flow alone to 1e-3 in test_waveglow_metric_width_vs_oracle.
#  * the feature extractors additionally carry the conditioning of a conv weight gradient in front of BatchNorm (dy is
#    orthogonal to 1 and x-hat: the sum cancels to ~1/600 of Σ|dy·x|; f32 itself measures 4e-5 there, split-bf16 2.6e-3).
_UPSTREAM_OF_RELU_HEADS = ("fe_t", "fe_s", "dimunif", "clf_t", "nf", "noise", "ad_net")
# The f32 mode is not immune either (one run in ~five flipped a unit at B=4: 3e-3 on clf_t.hidden.weight): fp32 atomics in
# the BatchNorm moment sums and the K-split random-layer GEMM move the last bits between runs of the SAME binary.  So the
# whole-step gate is: 1e-3 (f32) / 3e-3 (split-bf16) for everything not upstream of the ReLU heads, 5e-2 upstream — an
# indexing or layout bug shows as O(1) — and the tight gradient gates are the per-module tests (flow, extractor,
# classifier step, every kernel against fp64), which contain no such head.
GRAD_TOL = {"f32": dict({"default": 1e-3}, **{m: 5e-2 for m in _UPSTREAM_OF_RELU_HEADS}),
            "bf16x3": dict({"default": 3e-3}, **{m: 5e-2 for m in _UPSTREAM_OF_RELU_HEADS})}


@pytest.fixture(params=["bf16x3", "f32"])
def arithmetic(request):
    from feature_level_style_transfer_for_tsc_amd import ops
    prev, ops.MATH = ops.MATH, request.param
    yield request.param
    ops.MATH = prev


def _check_step(rep_o, want, rep, grads, tr, what, math="bf16x3"):
    for k in LOSSES:
        a, b = float(rep[k]), float(rep_o[k])
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (what, k, a, b)
    for k in ("logit_t", "logit_s", "logit_s2t"):
        close(rep[k], rep_o[k], 1e-4, f"{what} {k}")
    close(rep["feat_s2t"], rep_o["feat_s2t"], 1e-4, f"{what} feat_s2t")
    close(rep["norms_t"], rep_o["norms_t"], 1e-3, f"{what} GradNorm norms_t")
    close(rep["norms_s"], rep_o["norms_s"], 1e-3, f"{what} GradNorm norms_s")
    close(rep["w_t"], rep_o["w_t"], 1e-4, f"{what} w_t")
    close(rep["w_s"], rep_o["w_s"], 1e-4, f"{what} w_s")
    tol = GRAD_TOL[math]
    for name in tr.MODULES:
        check_grads(tr.m[name], want[name], tol.get(name, tol["default"]), f"{what} Q3 {name} ", grads=grads[name])


@pytest.mark.parametrize("L,B,seed", [(512, 4, 512), (1024, 2, 1024)])
def test_whole_joint_step_with_gradients_vs_oracle(L, B, seed, arithmetic):
    """configs[1] (L=512) and configs[4] (L=1024) geometry: forward, GradNorm and the accumulated gradients of the
    whole step, all eleven modules, against the oracle from identical seeded state."""
    js = R.build_joint_step(L, 1, L, 1, 4, 4, seed=seed, dropout_p=0.0, zero_end=False)
    tr = _trainer_from(js, L, L, 4)
    gen = torch.Generator().manual_seed(seed + 1)
    batch = (_pair(gen, B, 1, L, 4), _pair(gen, B, 1, L, 4))
    _check_step(*_step_both(js, tr, batch, (L // 8, L // 16)), tr, f"L={L} {arithmetic}", math=arithmetic)


def test_config2_four_sources_one_step_each_then_vote():
    """configs[2]: four source domains at len=512.  The reference trains one pipeline per source (main.py:7-11) and
    votes over the saved target-side checkpoints (multi_source_voting.py:265-277,281-424).  Here: four JointTrainers
    (different seeds, i.e. different source domains and initialisations) take one step each — every step checked
    against the oracle incl. gradients — then the four post-step (extractor, classifier) pairs vote on a test set:
    K=4 eval forward + batched vote on the device against the oracle's eval forward of the SAME post-step weights
    (copied back to the CPU: the first RMSprop step of a zero-gradient bias is +-10*lr by the sign of rounding noise,
    DESIGN.md "chaotic trajectory", so post-step weights are compared by what they predict, not bit for bit) fed to
    the restated voting block."""
    L, B, ncls, K = 512, 3, 4, 4
    gen = torch.Generator().manual_seed(2024)
    target = _pair(gen, B, 1, L, ncls)                                     # one target domain, four sources
    models, oracle_params = [], []
    for k in range(K):
        js = R.build_joint_step(L, 1, L, 1, ncls, ncls, seed=100 + k, dropout_p=0.0, zero_end=False)
        tr = _trainer_from(js, L, L, ncls)
        batch = (target, _pair(gen, B, 1, L, ncls))
        _check_step(*_step_both(js, tr, batch, (17 + k, 40 - k)), tr, f"source {k}")
        fe, clf = tr.m["fe_t"], tr.m["clf_t"]
        # one optimisation step leaves the BatchNorm running statistics at 0.9·init + 0.1·batch: eval-mode logits of such a
        # model are in the hundreds, where the reference's own vote — exp(x)/Σexp(x) in float32 (multi_source_voting.py:407)
        # — overflows to NaN.  Let the statistics settle (train-mode forwards, no parameter update) as an epoch would.
        with torch.no_grad():
            for _ in range(40):
                clf(fe(target[0].to(DEV)))
        fe.eval(); clf.eval()
        models.append((fe, clf))
        oracle_params.append(({n: v.detach().cpu() for n, v in fe.state_dict().items()},
                              {n: v.detach().cpu() for n, v in clf.state_dict().items()}))
    mk = lambda n: [_pair(gen, 5, 1, L, ncls) for _ in range(n)]
    train, test = mk(3), mk(2)
    w, scores, pred, acc = fst.multi_source_voting(models, train, test)
    fe_spec, clf_spec = R.train_specs(L, 1)

    def oracle_logits(batches):
        out = [torch.cat([R.classifier(R.feature_extractor(x, Pf, fe_spec, False), Pc, clf_spec, False)[0] for x, _ in batches])
               for Pf, Pc in oracle_params]
        return torch.stack(out).detach().numpy(), torch.cat([y for _, y in batches]).numpy()
    (trl, try_), (tel, tey) = oracle_logits(train), oracle_logits(test)
    got_tel, _ = fst.collect_logits(models, test)
    close(got_tel, tel, 1e-4, "K=4 eval logits")
    w_o, scores_o, pred_o, acc_o = R.multi_source_vote(trl, try_, tel, tey)
    np.testing.assert_allclose(w.cpu().numpy(), w_o, rtol=1e-9, atol=0)
    np.testing.assert_allclose(scores.cpu().numpy(), scores_o, rtol=2e-3)   # 9^w (1+120 e^-H) amplifies the 1e-4 logit tolerance
    assert np.array_equal(pred.cpu().numpy(), pred_o) and abs(acc - acc_o) < 1e-12


def test_full_batch_graph_replay_equals_eager_step():
    """B=256, L=512 (the bench configuration): one replay of the captured hipGraph equals the eager step from the
    same snapshot — nine losses, logits, GradNorm weights, and the post-step state of parameters with real gradients."""
    B, L = 256, 512
    torch.manual_seed(1234)
    tr = fst.JointTrainer(fst.JointConfig(L_t=L, L_s=L, dropout_p=0.0), DEV)  # the bench configuration, dropout off (its masks are drawn per call)
    gen = torch.Generator().manual_seed(7)
    (x_t, y_t), (x_s, y_s) = _pair(gen, B, 1, L, 4), _pair(gen, B, 1, L, 4)
    args = (x_t.to(DEV), y_t.to(DEV), x_s.to(DEV), y_s.to(DEV))
    tr.capture(*args, epoch=0)
    snap = tr.snapshot()
    rep = {k: v.clone() for k, v in tr.replay(*args, (31, 77)).items()}
    after_graph = tr.snapshot()
    tr.restore(snap)
    eager = tr.step(*args, epoch=0, t_samples=(31, 77))
    for k in LOSSES:
        a, b = float(rep[k]), float(eager[k])
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (k, a, b)
    # logit_s2t: the eval-mode classifier on the transferred features, after 11 optimisation steps from random
    # initialisation its running statistics are far from the batch's and the logits are ~1e8: fp32-atomic sum order (the
    # two runs differ in the last bits) is amplified to ~1e-3 there
    for k in ("logit_t", "logit_s", "logit_s2t", "w_t", "w_s", "norms_t", "norms_s"):
        close(rep[k], eager[k], 3e-3 if k == "logit_s2t" else (1e-4 if k.startswith("logit") or k.startswith("w_") else 1e-3),
              f"graph vs eager {k}")
    after_eager = tr.snapshot()
    # weights whose gradient is real (not rounding noise in front of a BatchNorm): RMSprop's first-step size is
    # lr*g/sqrt(0.01 g^2) = 10*lr whatever |g|, so equal signs give equal steps
    for k in ("m.fe_t.net_1.net.net.1.conv1d.weight", "m.nf.WN.0.in_layers.3.weight_v", "m.nf.WN.2.res_skip_layers.7.weight_g",
              "m.clf_t.hidden.weight", "m.cpc.Wk.0.weight", "w_s"):
        d = (after_graph["t"][k] - after_eager["t"][k]).abs()
        scale = float(after_eager["t"][k].abs().max())
        assert float((d > 2e-3 * scale).double().mean()) < 0.01, (k, float(d.max()), scale)
