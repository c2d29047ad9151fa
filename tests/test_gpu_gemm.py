"""fst_gemm (csrc/gemm.hip) — the dense products of the small heads on the matrix cores in split-bf16 — against fp64 on the
MI355X: every operand layout a Linear layer uses (y = x·Wᵀ, dx = g·W, dW = gᵀ·x), ragged shapes (rows and reduction lengths
that are not multiples of the tile, of 4 samples, a single output column), the K-split path and its fixed-order slab sum, bias
and activations in the epilogue; ``LinearActFn`` (widgets.py:32-42, :73-75, :113-131 of the reference: Linear + ReLU / LeakyReLU)
against the fp64 composition, output and every gradient; the heads built on it against their stock-torch twins.

Tolerance: 2e-5 of the output scale (split-bf16: three bf16 products, fp32 accumulate — ≈5e-6 of the scale measured)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import feature_level_style_transfer_for_tsc_amd as fst
from feature_level_style_transfer_for_tsc_amd import _lib, ops

DEV = "cuda"
needs_bf3 = pytest.mark.skipif(ops.MATH != "bf16x3", reason="fst_gemm is the split-bf16 path (FST_MATH=f32 keeps the library's exact-f32 GEMM)")


def close(got, want, tol, what):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    s = max(1e-6, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * s, f"{what}: max err {err:.3e}, scale {s:.3e}, tol {tol}"


def act64(v, act, slope, mask=None):
    """The activation in fp64; with ``mask`` (the branch each unit took on the device, out > 0) the reference is evaluated on the
    device's linear piece — a unit whose pre-activation is within rounding of zero may take either branch, and one such unit moves
    the gradients by percent — after checking that every disagreeing unit IS within rounding of zero."""
    if act == ops.ACT_NONE:
        return v
    if mask is None:
        mask = v > 0
    else:
        diff = (v.detach() > 0) != mask
        if bool(diff.any()):
            worst = float(v.detach()[diff].abs().max()) / max(1e-30, float(v.detach().abs().max()))
            assert worst <= 1e-4, f"{int(diff.sum())} unit(s) on the other branch with |x|/max|x| up to {worst:.2e}"
    return torch.where(mask, v, (slope if act == ops.ACT_LEAKY else 0.0) * v)


SHAPES = [
    # M, N, K
    (256, 1024, 1024),      # the CDAN discriminator's hidden layers (K split)
    (1600, 512, 512),       # rows of DimensionUnification's length GEMM (128 x 128 tiles, no split)
    (256, 400, 800),        # FeatureDiscriminatorforSource 800 -> 400
    (256, 800, 50),         # ... 50 -> 800: K % 4 != 0, rows not 16-byte aligned
    (256, 1, 50),           # a single output column
    (37, 70, 45),           # nothing divides anything
    (512, 512, 6400),       # the weight-gradient shape of the length GEMM (long reduction, K split over 128 x 128 tiles)
    (3, 4, 200),            # classifier head
]


@needs_bf3
@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_layouts_vs_fp64(M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + 2 * ta + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(DEV)
    B = torch.randn((K, N) if tb else (N, K), generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    A64 = (A.t() if ta else A).double()
    B64 = (B.t() if tb else B).double()
    want = A64 @ B64.t()
    close(ops.gemm(A, ta, B, tb), want, 2e-5, f"gemm {M}x{N}x{K} ta={ta} tb={tb}")
    for act, slope in ((ops.ACT_RELU, 0.0), (ops.ACT_LEAKY, 0.2)):
        got = ops.gemm(A, ta, B, tb, bias, act, slope)
        close(got, act64(want + bias.double(), act, slope), 2e-5, f"gemm+bias+act{act} {M}x{N}x{K} ta={ta} tb={tb}")   # (values: either branch of a ~0 unit is ~0)
    # the same bits on every run (fixed-order slab sum, no atomics)
    assert torch.equal(ops.gemm(A, ta, B, tb, bias), ops.gemm(A, ta, B, tb, bias))


@needs_bf3
def test_gemm_row_pitch_and_views():
    """Operands that are column slices of wider matrices (row pitch > row length, rows not 16-byte aligned)."""
    g = torch.Generator().manual_seed(3)
    wide_a, wide_b = torch.randn(130, 77, generator=g).to(DEV), torch.randn(90, 77, generator=g).to(DEV)
    A, B = wide_a[:, 3:60], wide_b[:, 5:62]
    close(ops.gemm(A, False, B, False), A.double() @ B.double().t(), 2e-5, "pitched operands")
    A2, B2 = wide_a[:57, 1:71], wide_b[:57, 2:33]                       # [K, M] and [K, N] views
    close(ops.gemm(A2, True, B2, True), A2.double().t() @ B2.double(), 2e-5, "pitched k-major operands")


def test_gemm_refuses_what_it_does_not_serve():
    lib = _lib.load()
    a = torch.zeros(8, 8, device=DEV)
    assert lib.fst_gemm(a.data_ptr(), 4, 0, a.data_ptr(), 8, 0, a.data_ptr(), 8, 8, 8, 8, None, 0, 0.0, None, 0, None) != 0   # lda < K
    assert lib.fst_gemm(a.data_ptr(), 8, 0, a.data_ptr(), 8, 0, a.data_ptr(), 8, 8, 8, 8, None, 7, 0.0, None, 0, None) != 0   # activation
    big = lib.fst_gemm_workspace_floats(256, 1024, 1024)
    assert big > 0
    x = torch.zeros(256, 1024, device=DEV)
    w = torch.zeros(1024, 1024, device=DEV)
    y = torch.zeros(256, 1024, device=DEV)
    assert lib.fst_gemm(x.data_ptr(), 1024, 0, w.data_ptr(), 1024, 0, y.data_ptr(), 1024, 256, 1024, 1024, None, 0, 0.0, None, 0, None) != 0
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(4, 4), False, torch.zeros(4, 4), False)   # CPU tensors: no fallback


@needs_bf3
@pytest.mark.parametrize("act,slope", [(ops.ACT_NONE, 0.0), (ops.ACT_RELU, 0.0), (ops.ACT_LEAKY, 0.2)])
@pytest.mark.parametrize("lead,K,N", [((256,), 800, 400), ((9, 25), 96, 64), ((64, 50), 512, 512), ((256,), 50, 1)])
def test_linear_act_fn_vs_fp64(act, slope, lead, K, N):
    g = torch.Generator().manual_seed(K + N + act)
    x = torch.randn(*lead, K, generator=g).to(DEV).requires_grad_(True)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
    cot = torch.randn(*lead, N, generator=g).to(DEV)
    y = ops.LinearActFn.apply(x, W, b, act, slope)
    got = torch.autograd.grad(y, [x, W, b], cot)
    x64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    want = act64(F.linear(x64, W64, b64), act, slope, (y.detach() > 0).reshape(*lead, N))
    wg = torch.autograd.grad(want, [x64, W64, b64], cot.double())
    close(y, want, 2e-5, "y")
    for a, w_, name in zip(got, wg, ("dx", "dW", "db")):
        close(a, w_, 5e-5, name)
    with ops.partial_backward():                                        # GradNorm's partial passes: no parameter gradients
        y2 = ops.LinearActFn.apply(x, W, b, act, slope)
        dx2, = torch.autograd.grad(y2, [x], cot)
    assert torch.equal(dx2, got[0])


@needs_bf3
def test_heads_on_fst_gemm_vs_stock_modules():
    """FeatureDiscriminatorforSource and AdversarialNetworkforCDAN (GRL + MLP) forward and backward on fst_gemm against the same
    modules' stock nn.Sequential / nn.Linear evaluation in fp64 (dropout off: the masks are torch's either way)."""
    torch.manual_seed(11)
    fd = fst.FeatureDiscriminatorforSource(50).to(DEV)
    ad = fst.AdversarialNetworkforCDAN(1024, 1024).to(DEV)
    ad.dropout1.p = ad.dropout2.p = 0.0
    for mod, width in ((fd, 50), (ad, 1024)):
        x = torch.randn(256, width, device=DEV, requires_grad=True)
        it0 = mod.iter_num
        masks, apply0 = [], ops.LinearActFn.apply

        def recording(*a):
            out = apply0(*a)
            if a[3] != ops.ACT_NONE:
                masks.append(out.detach() > 0)
            return out
        ops.LinearActFn.apply = recording
        try:
            y = mod(x)
        finally:
            del ops.LinearActFn.apply
        assert len(masks) == (3 if mod is fd else 2)
        coeff = fst.cdan.calc_coeff(mod.iter_num, mod.high, mod.low, mod.alpha, mod.max_iter)
        got = torch.autograd.grad(y.sum(), [x] + list(mod.parameters()))
        m64 = {k: v.detach().double().requires_grad_(True) for k, v in mod.named_parameters()}
        x64 = x.detach().double().requires_grad_(True)
        if mod is fd:
            h = x64
            for j, i in enumerate((0, 2, 4)):
                h = act64(F.linear(h, m64[f"model.{i}.weight"], m64[f"model.{i}.bias"]), ops.ACT_LEAKY, 0.2, masks[j])
            want = F.linear(h, m64["model.6.weight"], m64["model.6.bias"])
        else:
            h = act64(F.linear(x64, m64["ad_layer1.weight"], m64["ad_layer1.bias"]), ops.ACT_RELU, 0.0, masks[0])
            h = act64(F.linear(h, m64["ad_layer2.weight"], m64["ad_layer2.bias"]), ops.ACT_RELU, 0.0, masks[1])
            want = F.linear(h, m64["ad_layer3.weight"], m64["ad_layer3.bias"])
        wg = torch.autograd.grad(want.sum(), [x64] + [m64[k] for k, _ in mod.named_parameters()])
        close(y, want, 5e-5, "head output")
        close(got[0], -coeff * wg[0], 2e-4, "head dx (through the gradient reversal)")
        for a, w_, (name, _) in zip(got[1:], wg[1:], mod.named_parameters()):
            close(a, w_, 2e-4, name)
        assert mod.iter_num == it0 + 1
