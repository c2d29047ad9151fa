"""CDAN() host logic that needs no GPU: the quirk-Q4 product form and the no-random-layer branch (the outer product
p ⊗ f fed to the critic, C_DAN.py:57-61) against the reference's formulas written out with plain torch ops."""
import torch

from feature_level_style_transfer_for_tsc_amd.cdan import CDAN, Entropy
from feature_level_style_transfer_for_tsc_amd.widgets import AdversarialNetworkforCDAN


def _reference_formula(f_t, f_g, logit_t, logit_g, ad_net, coeff):
    """C_DAN.py:49-82 with random_layer=None, literally (bmm, view, hooks, the [B]·[B,1] broadcast)."""
    f_t, f_g = torch.flatten(f_t, 1), torch.flatten(f_g, 1)
    p_t, p_g = torch.softmax(logit_t, 1), torch.softmax(logit_g, 1)
    fus_t = torch.bmm(p_t.unsqueeze(2), f_t.unsqueeze(1))
    out_t = ad_net(fus_t.view(-1, f_t.size(1) * p_t.size(1)))
    fus_g = torch.bmm(p_g.unsqueeze(2), f_g.unsqueeze(1))
    out_g = ad_net(fus_g.view(-1, f_g.size(1) * p_g.size(1)))
    e_t, e_g = Entropy(p_t), Entropy(p_g)
    coeff = ad_net.coeff if coeff is None else coeff            # read once, after BOTH critic calls (C_DAN.py:69)
    e_t.register_hook(lambda g: -coeff * g.clone())
    e_g.register_hook(lambda g: -coeff * g.clone())
    w_t, w_g = 1.0 + torch.exp(-e_t), 1.0 + torch.exp(-e_g)
    w_t = w_t / torch.sum(w_t).detach().item()
    w_g = w_g / torch.sum(w_g).detach().item()
    return torch.sum(w_t * out_t) - torch.sum(w_g * out_g)


def test_cdan_without_random_layer_matches_the_reference_formula():
    torch.manual_seed(0)
    B, C, L, ncls = 5, 3, 4, 3
    ad = AdversarialNetworkforCDAN(C * L * ncls, 16)
    ad.dropout1.p = ad.dropout2.p = 0.0
    ad.eval()                                                   # the GRL counter does not advance: both calls see one coeff
    leaves = [torch.randn(B, C, L, requires_grad=True), torch.randn(B, C, L, requires_grad=True),
              torch.randn(B, ncls, requires_grad=True), torch.randn(B, ncls, requires_grad=True)]
    got = CDAN(*leaves, ad, None)
    g_got = torch.autograd.grad(got, leaves + list(ad.parameters()))
    want = _reference_formula(*leaves, ad, ad.coeff)
    g_want = torch.autograd.grad(want, leaves + list(ad.parameters()))
    assert abs(float(got) - float(want)) <= 1e-6 * max(1.0, abs(float(want)))
    for a, b in zip(g_got, g_want):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_cdan_reads_the_grl_coefficient_after_both_critic_calls():
    """Train mode: the critic's coefficient advances with every forward call (0 on call one, 0.987 on call two of the
    first batch); both entropy gradients are reversed with the value read after the second call (C_DAN.py:62-72)."""
    import copy
    torch.manual_seed(1)
    B, C, L, ncls = 4, 2, 5, 3
    ad = AdversarialNetworkforCDAN(C * L * ncls, 16)
    ad.dropout1.p = ad.dropout2.p = 0.0
    ad.train()
    twin = copy.deepcopy(ad)
    leaves = [torch.randn(B, C, L, requires_grad=True), torch.randn(B, C, L, requires_grad=True),
              torch.randn(B, ncls, requires_grad=True), torch.randn(B, ncls, requires_grad=True)]
    got = CDAN(*leaves, ad, None)
    g_got = torch.autograd.grad(got, leaves)
    want = _reference_formula(*leaves, twin, None)
    g_want = torch.autograd.grad(want, leaves)
    assert ad.coeff == twin.coeff and ad.coeff > 0.9
    for a, b in zip(g_got, g_want):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
