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
    q.put((rank, params[0].grad.clone(), params[1].grad.clone(), params[2].grad, s))
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
