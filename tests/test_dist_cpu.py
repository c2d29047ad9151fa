"""N>1 path on the CPU: two gloo ranks, one flat gradient bucket."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from feature_level_style_transfer_for_tsc_amd.dist import GradBucket, shard_batch


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    params[0].grad = torch.full((3, 5), float(rank + 1))
    params[1].grad = torch.arange(7.0) * (rank + 1)
    bucket = GradBucket()
    bucket.all_reduce(params)                                     # params[2] has no grad: skipped
    s = bucket.mean_scalars(torch.tensor([float(rank), 10.0 * rank]))
    # plain numpy over the queue: a tensor travels as a file descriptor the parent must fetch while this process is still
    # alive (torch.multiprocessing's fd sharing) — a worker that exits first resets the connection
    q.put((rank, params[0].grad.numpy().copy(), params[1].grad.numpy().copy(), params[2].grad, s.numpy().copy()))
    dist.destroy_process_group()


def test_grad_bucket_two_ranks():
    world, port = 2, 29611
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, g0, g1, g2, s in out:
        g0, g1, s = torch.from_numpy(g0), torch.from_numpy(g1), torch.from_numpy(s)
        assert torch.allclose(g0, torch.full((3, 5), 1.5))
        assert torch.allclose(g1, torch.arange(7.0) * 1.5)
        assert g2 is None
        assert torch.allclose(s, torch.tensor([0.5, 5.0]))


def test_shard_batch_covers_everything():
    for n, world in ((2048, 8), (10, 4), (3, 8)):
        seen = []
        for r in range(world):
            sl = shard_batch(n, r, world)
            seen.extend(range(n)[sl])
        assert seen == list(range(n))


# --------------------------------------------------------------------------------------------------
# mode B ("global-batch exact"): the differentiable collectives
# --------------------------------------------------------------------------------------------------
def _worker_global(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from feature_level_style_transfer_for_tsc_amd import dist as D
    bucket = GradBucket()
    # global data: x_g [4, 3]; rank r owns rows [2r, 2r+2).  Loss per rank = mean over its rows of f(row; global stats)
    xg = torch.arange(12.0).view(4, 3) / 7.0
    x = xg[2 * rank: 2 * rank + 2].clone().requires_grad_(True)
    assert not D.global_batch_active() and D.world() == 1 and D.mean_over_ranks(x) is x and D.gather_cat(x, 0) is x
    with D.global_batch(bucket):
        assert D.global_batch_active() and D.world() == world and D.rank() == rank
        m = D.mean_over_ranks(x.mean(dim=0))                     # global batch mean [3]
        allx = D.gather_cat(x, 0)                                # [4, 3] in rank order
        s = torch.ones(2)
        n = D.sum_over_ranks_(s)
        # local loss: mean over local rows of ((x - m)^2).sum() + coupling of local rows with ALL rows
        loss = ((x - m) ** 2).sum(dim=1).mean() + (x @ allx.t()).sum(dim=1).mean() / 4.0
        loss.backward()
    q.put((rank, m.detach().numpy().copy(), allx.detach().numpy().copy(), s.numpy().copy(), n, float(loss), x.grad.numpy().copy()))
    dist.destroy_process_group()


def test_global_batch_collectives_match_single_process():
    world, port = 2, 29613
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_global, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process reference on the concatenated batch: L = mean over ALL rows of the same per-row loss
    xg = (torch.arange(12.0).view(4, 3) / 7.0).requires_grad_(True)
    m = xg.mean(dim=0)
    L = ((xg - m) ** 2).sum(dim=1).mean() + (xg @ xg.t()).sum(dim=1).mean() / 4.0
    L.backward()
    losses = []
    for rank, m_r, allx, s, n, loss, gx in out:
        m_r, allx, s, gx = (torch.from_numpy(v) for v in (m_r, allx, s, gx))
        assert torch.allclose(m_r, m.detach()) and torch.allclose(allx, xg.detach())
        assert n == world and torch.allclose(s, torch.full((2,), float(world)))
        losses.append(loss)
        # every rank differentiates ITS mean; averaging parameter gradients over ranks (the bucket) then equals the
        # gradient of the global mean — for a leaf input that is: global grad = (1/world) · local grad
        assert torch.allclose(gx / world, xg.grad[2 * rank: 2 * rank + 2], atol=1e-6), (rank, gx / world, xg.grad)
    assert abs(sum(losses) / world - float(L)) < 1e-6
