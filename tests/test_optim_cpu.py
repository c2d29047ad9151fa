"""SharedStepAdam == torch.optim.Adam (same update formula, one shared step counter) on a bag of small tensors that all
receive a gradient every step — the only way the hot path uses it (a tensor that skipped a step would keep its own
counter in torch's Adam, which is exactly the per-tensor state this optimiser does away with)."""
import torch

from feature_level_style_transfer_for_tsc_amd.optim import SharedStepAdam


def test_matches_torch_adam_over_several_steps():
    g = torch.Generator().manual_seed(0)
    shapes = [(5, 3), (7,), (2, 4, 3), (1,)] * 5
    a = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref, mine = torch.optim.Adam(a, lr=0.002), SharedStepAdam(b, lr=0.002)
    for step in range(7):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g) * (10.0 ** (step % 3 - 1))
            pa.grad, pb.grad = gr.clone(), gr.clone()
        ref.step(); mine.step()
        for pa, pb in zip(a, b):
            assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-7), step
    assert float(mine.param_groups[0]["step"]) == 7.0
    assert mine.state[b[0]]["step"] is mine.param_groups[0]["step"]  # exposed for state snapshots
