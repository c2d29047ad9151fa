"""Host logic of the conv engine, checked on the CPU: a numpy emulation of the device index math
(plan.emulate_pack / emulate_conv) must reproduce F.conv1d for every plan shape the modules build."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from feature_level_style_transfer_for_tsc_amd import ops
from feature_level_style_transfer_for_tsc_amd.plan import HDR, WG_ITEMS, Segment, build_plan, emulate_conv, emulate_pack
from feature_level_style_transfer_for_tsc_amd.structure import (generate_layer_parameter_list, out_channels,
                                                                 row_live_ranges)


def _ref_conv(x, w, dil, pad_left, ntaps):
    L = x.shape[-1]
    halo = (ntaps - 1) * dil
    xp = F.pad(x.double(), (pad_left, halo - pad_left))
    return F.conv1d(xp[None], w.double(), dilation=dil)[0, :, :L]


CASES = [  # M, C0, ntaps, dil, pad_left, C1
    (8, 6, 1, 1, 0, 0), (50, 50, 1, 1, 0, 0), (16, 8, 3, 4, 4, 3), (40, 5, 3, 1, 1, 0), (70, 33, 2, 1, 0, 0)]


@pytest.mark.parametrize("M,C0,ntaps,dil,pad_left,C1", CASES)
def test_forward_and_dx_plans_reproduce_conv1d(M, C0, ntaps, dil, pad_left, C1):
    g = torch.Generator().manual_seed(M * 131 + C0)
    L = 37
    spec = ops.ConvSpec(M, C0, ntaps, dil, pad_left, C1=C1)
    w0 = torch.randn(M, C0, ntaps, generator=g)
    x0 = torch.randn(C0, L, generator=g)
    w1 = torch.randn(M, C1, 1, generator=g) if C1 else None
    x1 = torch.randn(C1, L, generator=g) if C1 else None
    want = _ref_conv(x0, w0, dil, pad_left, ntaps)
    if C1:
        want = want + _ref_conv(x1, w1, 1, 0, 1)
    for nb in (1, 2):
        plan = spec.fwd_plan(nb)
        a = emulate_pack(plan, [w0.numpy(), w1.numpy() if C1 else None], [spec.s_w0(), spec.s_w1()])
        got = emulate_conv(plan, a, [x0.numpy(), x1.numpy() if C1 else None], L)
        np.testing.assert_allclose(got, want.numpy(), rtol=1e-5, atol=1e-5)
    # data gradient of the main input = conv of dy with the transposed, tap-flipped weights
    dy = torch.randn(M, L, generator=g)
    x0r = x0.clone().double().requires_grad_(True)
    (_ref_conv_autograd(x0r, w0, dil, pad_left, ntaps) * dy.double()).sum().backward()
    plan = spec.dx0_plan(1)
    a = emulate_pack(plan, [w0.numpy(), None], [spec.s_w0_T(), (0, 0, 0, 0)])
    got = emulate_conv(plan, a, [dy.numpy(), None], L)
    np.testing.assert_allclose(got, x0r.grad.numpy(), rtol=1e-5, atol=1e-5)


def _ref_conv_autograd(x, w, dil, pad_left, ntaps):
    L = x.shape[-1]
    halo = (ntaps - 1) * dil
    return F.conv1d(F.pad(x, (pad_left, halo - pad_left))[None], w.double(), dilation=dil)[0, :, :L]


@pytest.mark.parametrize("end,budgets,cin", [(6, [44, 700], 2), (13, [41 * 3, 41 * 12 * 2], 1)])
def test_omni_scale_plans_skip_masked_taps(end, budgets, cin):
    lp = generate_layer_parameter_list(1, end, budgets, cin)
    g = torch.Generator().manual_seed(5)
    for layer in lp:
        C0, kmax, M = layer[0][0], layer[-1][2], out_channels(layer)
        live = row_live_ranges(layer)
        mask = torch.zeros(M, C0, kmax)
        for m, (lo, hi) in enumerate(live):
            mask[m, :, lo:hi] = 1
        w = torch.randn(M, C0, kmax, generator=g) * mask
        x = torch.randn(C0, 29, generator=g)
        pl = int((kmax - 1) / 2)
        spec = ops.ConvSpec(M, C0, kmax, 1, pl, row_live=live)
        plan = spec.fwd_plan(1)
        a = emulate_pack(plan, [w.numpy(), None], [spec.s_w0(), (0, 0, 0, 0)])
        np.testing.assert_allclose(emulate_conv(plan, a, [x.numpy(), None], 29), _ref_conv(x, w, 1, pl, kmax).numpy(),
                                   rtol=1e-5, atol=1e-5)
        if kmax > 2 and plan.n_mgroups > 1:
            assert plan.total_records * plan.MB * 32 < n_mgroup_rows(plan) * ((C0 + 1) // 2) * kmax   # fewer than dense
        dy = torch.randn(M, 29, generator=g)
        xr = x.clone().double().requires_grad_(True)
        (_ref_conv_autograd(xr, w, 1, pl, kmax) * dy.double()).sum().backward()
        dplan = spec.dx0_plan(1)
        a = emulate_pack(dplan, [w.numpy(), None], [spec.s_w0_T(), (0, 0, 0, 0)])
        np.testing.assert_allclose(emulate_conv(dplan, a, [dy.numpy(), None], 29), xr.grad.numpy(), rtol=1e-5, atol=1e-5)


def n_mgroup_rows(plan):
    return plan.n_mgroups * plan.MB * 32


def test_metric_config_plan_shapes():
    """L=512, C_in=1: layer 1 (25→225, primes 1..89) multiplies 216 900 live MACs/timestep out of 500 625 dense."""
    lp = generate_layer_parameter_list(1, 89, [1024, 229376], 1)
    layer = lp[1]
    live = row_live_ranges(layer)
    assert sum((hi - lo) * 25 for lo, hi in live) == 216900
    prev = ops.MATH
    try:
        ops.MATH = "f32"                                            # f32 window kernel: one 26-channel window
        spec = ops.ConvSpec(225, 25, 89, 1, 44, row_live=live)
        plan = spec.fwd_plan(2)
        assert plan.MB == 1 and plan.n_mgroups == 8 and plan.n_chunks == 1 and not plan.windowed16
        packed_macs = plan.total_records * 2 * 32                  # MACs/timestep the kernel issues (incl. padding)
        assert 216900 <= packed_macs <= 1.7 * 216900
        ops.MATH = "bf16x3"                                         # bf16 window kernel: 16-channel chunks, 16-deep k-steps
        spec = ops.ConvSpec(225, 25, 89, 1, 44, row_live=live)
        plan = spec.fwd_plan(2)
        assert plan.MB == 1 and plan.n_mgroups == 8 and plan.n_chunks == 2 and plan.windowed16 and not plan.pipeable
        assert ops.bf3_ok(plan, 150) and ops.bf3_ok(plan, 512)      # no alignment requirement on L
        issued = plan.n_stages * 16 * 32                            # MACs/timestep incl. the 25 -> 32 channel padding
        assert 216900 <= issued <= 2.2 * 216900
        assert plan.packed_floats_bf3 == plan.n_stages * 512 + 4
    finally:
        ops.MATH = prev
    wg = spec.wg_plan()
    assert wg.MB == 8 and len(wg.items()) % WG_ITEMS == 0
    ent = wg.entries()
    assert (ent[:, :, 0] == 0).all() and (ent[:, :, 1] == 89).all()  # dense over Kmax (quirk Q1)


def test_item_table_groups_share_an_m_group():
    spec = ops.ConvSpec(240, 120, 3, 8, 8, C1=25)
    plan = spec.wg_plan()
    items = plan.items()
    assert len(items) and len(items) % WG_ITEMS == 0
    for w in range(len(items) // WG_ITEMS):
        gs = {int(it[0]) for it in items[w * WG_ITEMS:(w + 1) * WG_ITEMS] if it[1] >= 0}
        assert len(gs) == 1
    covered = sum(1 for it in items if it[1] >= 0)
    ent, ch = plan.entries(), plan.chunks()
    want = sum(((e[1] - e[0]) * ((ch[q][2] + 1) & ~1) + 31) // 32 for g in range(plan.n_mgroups)
               for q, e in enumerate(ent[g]) if e[1] > e[0])
    assert covered == want
    assert plan.table[8] == len(items) and plan.table[9] == WG_ITEMS and plan.length == plan.table.size


@pytest.mark.parametrize("h,n,nl", [(25, 120, 8), (3, 8, 2), (4, 16, 1)])
def test_wn_flat_weight_layout(h, n, nl):
    """WNSpecs.flatten / unflatten: the order and shapes WNFn documents, a lossless round trip of a real WN's folded
    weights, and the two adjacencies its backward relies on — the in_layer biases form one [nl·2n] run (copied into the
    cond_layer bias gradient) and the res_skip biases of layers 0..nl-2 one [nl-1, 2n] block."""
    import feature_level_style_transfer_for_tsc_amd as fst
    torch.manual_seed(n)
    wn = fst.WN(h, nl, n, 3)
    S = wn.specs
    ws = [w.detach() for w in wn._fold()]
    assert [tuple(w.shape) for w in ws] == list(S.shapes) and len(S.shapes) == 6 + 4 * nl
    flat = S.flatten(ws)
    assert flat.numel() == S.flat_numel == sum(w.numel() for w in ws)
    back = S.unflatten(flat)
    assert all(torch.equal(a, b) and a.data_ptr() == flat.data_ptr() + 4 * S.offsets[i] for i, (a, b) in enumerate(zip(back, ws)))
    in_b0, rs_b0 = 6 + nl, 6 + 3 * nl
    assert all(S.offsets[in_b0 + i + 1] - S.offsets[in_b0 + i] == 2 * n for i in range(nl))
    assert S.shapes[3] == (2 * n * nl,) and S.offsets[in_b0 + nl] - S.offsets[in_b0] == 2 * n * nl
    assert all(S.shapes[rs_b0 + i] == (2 * n,) for i in range(nl - 1)) and S.shapes[rs_b0 + nl - 1] == (n,)
