"""Data-parallel train step on the GPU box: two ranks share cuda:0 over the gloo backend (RCCL needs one GPU per
rank; the collectives the step issues are the same).  Checks that replicas stay bit-identical with different
per-rank batches, eagerly and through the two-graph hipGraph path (collectives between the graphs)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _guarded(fn):
    """A worker that dies with an exception would leave the parent waiting on the queue: report it instead."""
    def run(rank, world, port, q):
        try:
            fn(rank, world, port, q)
        except BaseException:                                              # noqa: BLE001 — forwarded to the parent
            import traceback
            q.put((rank, "worker failed", traceback.format_exc()))
            raise
    run.__name__ = fn.__name__
    return run


def _collect(q, world, procs, timeout=300):
    out = []
    for _ in range(world):
        item = q.get(timeout=timeout)
        assert not (len(item) == 3 and item[1] == "worker failed"), f"rank {item[0]}:\n{item[2]}"
        out.append(item)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return sorted(out, key=lambda t: t[0])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import feature_level_style_transfer_for_tsc_amd as fst
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(GOLDEN, "joint_small.npz"), allow_pickle=False))
    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    sub = lambda prefix: {k[len(prefix):]: torch.tensor(v) for k, v in g.items() if k.startswith(prefix)}
    cfg = fst.JointConfig(L_t=meta["L_t"], C_in_t=meta["C_in_t"], L_s=meta["L_s"], C_in_s=meta["C_in_s"],
                          n_class_t=meta["ncls_t"], n_class_s=meta["ncls_s"], nf_channels=meta["nf"][2],
                          cpc_hidden=meta["cpc"][1], cdan_dim=64, ad_hidden=32, dropout_p=0.0)
    tr = fst.JointTrainer(cfg, dev, fst.GradBucket(), fe_t_spec=tup(meta["lp_t"]), clf_spec=tup(meta["lp_clf"]),
                          fe_s_spec=tup(meta["lp_s"]))
    tr.load_params({name: sub(f"sd0.{name}.") for name in tr.MODULES}, [torch.tensor(g["m0"]), torch.tensor(g["m1"])])
    gen = torch.Generator().manual_seed(100 + rank)                       # different data on every rank
    B = meta["B"]
    mk = lambda C, L, n: (torch.randn(B, C, L, generator=gen).to(dev), torch.randint(n, (B,), generator=gen).to(dev))
    (x_t, y_t), (x_s, y_s) = mk(meta["C_in_t"], meta["L_t"], meta["ncls_t"]), mk(meta["C_in_s"], meta["L_s"], meta["ncls_s"])
    for step in range(2):
        tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(2, 5))
    digest_eager = torch.cat([p.detach().flatten() for p in tr.parameters()]).double().sum().item()
    torch.manual_seed(9)                                                  # same CPC draws on both ranks
    tr.capture(x_t, y_t, x_s, y_s, epoch=0)
    rep = tr.replay(x_t, y_t, x_s, y_s, (1, 3))
    torch.cuda.synchronize()
    digest_graph = torch.cat([p.detach().flatten() for p in tr.parameters()]).double().sum().item()
    q.put((rank, digest_eager, digest_graph, len(tr._graphs), float(rep["w_t"].sum()), float(rep["nf_t"])))
    dist.destroy_process_group()


def _run_worker(name, rank, world, port, q):
    _guarded(globals()[name])(rank, world, port, q)


def test_two_rank_replicas_stay_identical():
    world, port = 2, 29633
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_worker, args=("_worker", r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = _collect(q, world, procs)
    (_, e0, g0, n0, w0, l0), (_, e1, g1, n1, w1, l1) = out
    assert e0 == e1, "replicas diverged in eager DP steps"
    assert g0 == g1, "replicas diverged through the captured DP step"
    assert n0 == n1 == 3                                                 # three graphs with the collectives between
    assert abs(w0 - 7.0) < 1e-4 and w0 == w1                             # GradNorm weights renormalised, in sync
    assert l0 != l1                                                      # ranks really saw different batches


# --------------------------------------------------------------------------------------------------
# mode B ("global-batch exact"): 2 ranks x B/2 samples == 1 process x B samples
# --------------------------------------------------------------------------------------------------
def _build_small_trainer(fst, dev, bucket, sync):
    g = dict(np.load(os.path.join(GOLDEN, "joint_small.npz"), allow_pickle=False))
    meta = json.loads(str(g["meta"]))
    tup = lambda lp: [[tuple(t) for t in l] for l in lp]
    sub = lambda prefix: {k[len(prefix):]: torch.tensor(v) for k, v in g.items() if k.startswith(prefix)}
    cfg = fst.JointConfig(L_t=meta["L_t"], C_in_t=meta["C_in_t"], L_s=meta["L_s"], C_in_s=meta["C_in_s"],
                          n_class_t=meta["ncls_t"], n_class_s=meta["ncls_s"], nf_channels=meta["nf"][2],
                          cpc_hidden=meta["cpc"][1], cdan_dim=64, ad_hidden=32, dropout_p=0.0)
    tr = fst.JointTrainer(cfg, dev, bucket, fe_t_spec=tup(meta["lp_t"]), clf_spec=tup(meta["lp_clf"]),
                          fe_s_spec=tup(meta["lp_s"]), sync=sync)
    tr.load_params({name: sub(f"sd0.{name}.") for name in tr.MODULES}, [torch.tensor(g["m0"]), torch.tensor(g["m1"])])
    return tr, meta


def _global_data(meta, n):
    gen = torch.Generator().manual_seed(4242)
    mk = lambda C, L, k: (torch.randn(n, C, L, generator=gen), torch.randint(k, (n,), generator=gen))
    return mk(meta["C_in_t"], meta["L_t"], meta["ncls_t"]), mk(meta["C_in_s"], meta["L_s"], meta["ncls_s"])


def _one_step_with_grads(tr, x_t, y_t, x_s, y_s):
    grads = {}
    tr.on_grads_ready = lambda: grads.update(
        {f"{k}.{n}": p.grad.detach().float().cpu().clone() for k in tr.MODULES for n, p in tr.m[k].named_parameters()
         if p.grad is not None})
    rep = tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(2, 5))
    torch.cuda.synchronize()
    losses = {k: float(rep[k]) for k in ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s", "cdan", "ce_s2t2s", "fd_s")}
    return losses, rep["logit_t"].float().cpu(), grads, rep["w_t"].float().cpu(), rep["w_s"].float().cpu()


def _worker_global(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import feature_level_style_transfer_for_tsc_amd as fst
    dev = torch.device("cuda:0")
    tr, meta = _build_small_trainer(fst, dev, fst.GradBucket(), "global")
    B = meta["B"]                                                         # per rank; the global batch is world*B
    (x_t, y_t), (x_s, y_s) = _global_data(meta, world * B)
    sl = slice(rank * B, (rank + 1) * B)
    losses, logit_t, grads, w_t, w_s = _one_step_with_grads(tr, x_t[sl].to(dev), y_t[sl].to(dev), x_s[sl].to(dev), y_s[sl].to(dev))
    q.put((rank, losses, logit_t.numpy(), {k: v.numpy() for k, v in grads.items()}, w_t.numpy(), w_s.numpy()))
    dist.destroy_process_group()


def test_global_batch_mode_reproduces_the_single_process_step():
    """SyncBN + CPC negatives from every rank + global NoiseTransfer means + CDAN's global batch sums + GradNorm on
    the averaged per-loss gradients: two ranks with half the batch each give the losses, logits, accumulated
    gradients (after the bucket) and GradNorm weights of one process on the whole batch."""
    world, port = 2, 29641
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_worker, args=("_worker_global", r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = _collect(q, world, procs)

    import feature_level_style_transfer_for_tsc_amd as fst
    dev = torch.device("cuda:0")
    tr, meta = _build_small_trainer(fst, dev, None, "ddp")
    B = meta["B"]
    (x_t, y_t), (x_s, y_s) = _global_data(meta, world * B)
    ref_losses, ref_logits, ref_grads, ref_w_t, ref_w_s = _one_step_with_grads(tr, x_t.to(dev), y_t.to(dev), x_s.to(dev), y_s.to(dev))

    # per-sample means: the global loss is the mean of the ranks' local losses; batch sums / global quantities are
    # reported identically by every rank
    local_mean = ("nf_t", "nf_s", "ce_t", "ce_s", "ce_s2t2s", "sl_t", "sl_s", "fd_s")
    for k in local_mean:
        got = sum(o[1][k] for o in out) / world
        assert abs(got - ref_losses[k]) <= 1e-4 * max(1.0, abs(ref_losses[k])), (k, got, ref_losses[k])
    for o in out:
        assert abs(o[1]["cdan"] - ref_losses["cdan"]) <= 1e-4 * max(1.0, abs(ref_losses["cdan"])), (o[1]["cdan"], ref_losses["cdan"])
    for rank, _, logit_t, grads, w_t, w_s in out:
        want = ref_logits[rank * B: (rank + 1) * B].numpy()
        assert np.abs(logit_t - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), "logits (SyncBN) differ"
        assert np.abs(w_t - ref_w_t.numpy()).max() <= 1e-4 and np.abs(w_s - ref_w_s.numpy()).max() <= 1e-4, "GradNorm weights"
        assert set(grads) == set(ref_grads)
        by_mod = {}
        for k, gr in grads.items():
            mod = k.split(".")[0]
            by_mod.setdefault(mod, [0.0, 0.0])
            ref = ref_grads[k].numpy()
            by_mod[mod][0] = max(by_mod[mod][0], float(np.abs(gr - ref).max()))
            by_mod[mod][1] = max(by_mod[mod][1], float(np.abs(ref).max()))
        for mod, (err, scale) in by_mod.items():
            assert err <= 2e-3 * max(scale, 1e-6), f"rank {rank}: gradients of {mod} differ: {err:.3e} vs scale {scale:.3e}"


# configs[4] (8 x MI355X, global batch 2048, L=1024): its per-rank geometry — the metric-size network at L=1024, 256 samples per
# rank — has to have gone through mode B's collectives once: SyncBN moment slots over 2 x 256 x 1024 samples, CPC scored against
# T=512 steps x 512 gathered predictions (two column panels), NoiseTransfer's [50, 1024] means, the 35.6 MB gradient bucket.
CFG4_L, CFG4_B = 1024, 256


def _build_config4_trainer(fst, dev, bucket, sync):
    torch.manual_seed(4096)                                                # same initial state in every process
    cfg = fst.JointConfig(L_t=CFG4_L, L_s=CFG4_L, dropout_p=0.0, nf_end_std=0.05)
    return fst.JointTrainer(cfg, dev, bucket, sync=sync)


def _config4_data(n):
    gen = torch.Generator().manual_seed(777)

    def mk():
        x = torch.randn(n, 1, CFG4_L, generator=gen)
        return (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True), torch.randint(4, (n,), generator=gen)
    return mk(), mk()


def _step_summary(tr, x_t, y_t, x_s, y_s):
    """losses, logits, GradNorm weights and per-module (max |grad|, a fixed random projection of the gradients) of one step:
    enough to compare two runs without shipping 35 MB of gradients through the queue."""
    summ = {}

    def grab():
        gen = torch.Generator(device=x_t.device).manual_seed(5)
        for k in tr.MODULES:
            gs = [p.grad.detach().flatten().double() for p in tr.m[k].parameters() if p.grad is not None]
            if gs:
                flat = torch.cat(gs)
                r = torch.randn(flat.numel(), 8, generator=gen, device=flat.device, dtype=torch.float64)
                summ[k] = (float(flat.abs().max()), (flat @ r / flat.numel() ** 0.5).cpu().numpy())
    tr.on_grads_ready = grab
    rep = tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(200, 75))
    torch.cuda.synchronize()
    losses = {k: float(rep[k]) for k in ("nf_t", "nf_s", "ce_t", "sl_t", "ce_s", "sl_s", "cdan", "ce_s2t2s", "fd_s")}
    return losses, rep["logit_t"].float().cpu().numpy(), summ, rep["w_t"].float().cpu().numpy(), rep["w_s"].float().cpu().numpy()


def _worker_global_config4(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import feature_level_style_transfer_for_tsc_amd as fst
    dev = torch.device("cuda:0")
    tr = _build_config4_trainer(fst, dev, fst.GradBucket(), "global")
    (x_t, y_t), (x_s, y_s) = _config4_data(world * CFG4_B)
    sl = slice(rank * CFG4_B, (rank + 1) * CFG4_B)
    q.put((rank,) + _step_summary(tr, x_t[sl].to(dev), y_t[sl].to(dev), x_s[sl].to(dev), y_s[sl].to(dev)))
    dist.destroy_process_group()


def test_global_batch_mode_at_config4_per_rank_geometry():
    """Mode B at configs[4]'s per-rank shape (L=1024, 256 samples per rank, the metric-size network): two ranks reproduce the
    single-process step on the 512-sample batch — nine losses, the SyncBN'd logits, GradNorm's weights and every module's
    accumulated gradient (compared through its largest entry's scale and a fixed 8-column random projection)."""
    world, port = 2, 29645
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_worker, args=("_worker_global_config4", r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = _collect(q, world, procs, timeout=600)

    import feature_level_style_transfer_for_tsc_amd as fst
    dev = torch.device("cuda:0")
    tr = _build_config4_trainer(fst, dev, None, "ddp")
    (x_t, y_t), (x_s, y_s) = _config4_data(world * CFG4_B)
    ref_losses, ref_logits, ref_summ, ref_w_t, ref_w_s = _step_summary(tr, x_t.to(dev), y_t.to(dev), x_s.to(dev), y_s.to(dev))
    for k in ("nf_t", "nf_s", "ce_t", "ce_s", "ce_s2t2s", "sl_t", "sl_s", "fd_s"):
        got = sum(o[1][k] for o in out) / world
        assert abs(got - ref_losses[k]) <= 1e-4 * max(1.0, abs(ref_losses[k])), (k, got, ref_losses[k])
    for rank, losses, logit_t, summ, w_t, w_s in out:
        assert abs(losses["cdan"] - ref_losses["cdan"]) <= 1e-4 * max(1.0, abs(ref_losses["cdan"])), (losses["cdan"], ref_losses["cdan"])
        want = ref_logits[rank * CFG4_B: (rank + 1) * CFG4_B]
        assert np.abs(logit_t - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), "logits (SyncBN) differ"
        assert np.abs(w_t - ref_w_t).max() <= 1e-4 and np.abs(w_s - ref_w_s).max() <= 1e-4, "GradNorm weights"
        assert set(summ) == set(ref_summ)
        for mod, (scale, proj) in summ.items():
            ref_scale, ref_proj = ref_summ[mod]
            assert abs(scale - ref_scale) <= 2e-3 * max(ref_scale, 1e-6), (mod, scale, ref_scale)
            # a projection of N gradient entries on unit-variance columns, divided by sqrt(N): entries of the order of the rms
            assert np.abs(proj - ref_proj).max() <= 2e-3 * max(ref_scale, 1e-6), (mod, np.abs(proj - ref_proj).max(), ref_scale)


def test_bench_data_parallel_path_rehearsal():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), rehearsed on
    this one-GPU box: both ranks on cuda:0, gloo instead of RCCL.  Guards the N>1 bench path — two captured graphs with
    the eager all-reduce between them, barrier + MAX timing, the per-kernel timing step on every rank — and its JSON."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FST_BENCH_ONE_DEVICE="1", FST_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--length", "128"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                            # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 32
    assert d["mode"] == "graph" and d["dist_backend"] == "gloo" and d["scaling"] == "weak"
    assert d["value"] > 0 and d["steps"] == 2 and d["warmup"] == 1 and "roofline" in d and "cpu_baseline" not in d


def test_plain_bench_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher environment (how a user — or a driver without torchrun — calls it):
    bench.py starts torch.distributed.run itself as a child before touching the GPU, relays rank 0's single JSON line and
    exits with the child's code.  One-device rehearsal switches as above."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(FST_BENCH_ONE_DEVICE="1", FST_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
           "--length", "128"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["mode"] == "graph" and d["value"] > 0
    # a rank that fails must fail the command: --gpus 2 under a launcher environment that says otherwise
    bad = subprocess.run(cmd, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=root, capture_output=True, text=True,
                         timeout=300)
    assert bad.returncode != 0


def _worker_rccl(rank, world, port, q):
    """One rank, backend "nccl" (= RCCL on ROCm): communicator init, a raw all-reduce, the gradient bucket and the 10-scalar
    average issued for real (always_reduce), eagerly and between the two captured graphs of the data-parallel step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import feature_level_style_transfer_for_tsc_amd as fst
    t = torch.arange(4096, device=dev, dtype=torch.float32)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(4096, dtype=torch.float32))
    assert dist.get_backend() == "nccl"
    bucket = fst.GradBucket(always_reduce=True)
    tr, meta = _build_small_trainer(fst, dev, bucket, "ddp")
    ref, _ = _build_small_trainer(fst, dev, None, "ddp")
    (x_t, y_t), (x_s, y_s) = _global_data(meta, meta["B"])
    x_t, y_t, x_s, y_s = (v.to(dev) for v in (x_t, y_t, x_s, y_s))
    a = tr.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(2, 5))
    b = ref.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=(2, 5))
    diffs = {k: (float(a[k]), float(b[k])) for k in ("nf_t", "ce_t", "cdan", "sl_s")}
    # two trainers of identical state: the same launches on the same numbers (no atomics on this path), 1e-6 is rounding room only
    eager_same = all(abs(x - y) <= 1e-6 * max(1.0, abs(y)) for x, y in diffs.values()) or diffs
    torch.manual_seed(9)
    tr.capture(x_t, y_t, x_s, y_s, epoch=0)
    rep = tr.replay(x_t, y_t, x_s, y_s, (1, 3))
    torch.cuda.synchronize()
    q.put((0, eager_same, len(tr._graphs), float(rep["w_t"].sum()), bool(torch.isfinite(rep["nf_t"]))))
    dist.destroy_process_group()


def test_rccl_backend_runs_the_data_parallel_step_on_one_rank():
    """RCCL itself (backend "nccl") has to have executed at least once before the 8-GPU driver run: world size 1 on this
    box's single GPU, with the bucket forced to issue its collectives."""
    world, port = 1, 29677
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_worker, args=("_worker_rccl", 0, world, port, q))]
    procs[0].start()
    (_, eager_same, n_graphs, w_sum, finite), = _collect(q, world, procs)
    assert eager_same is True and n_graphs == 3 and abs(w_sum - 7.0) < 1e-4 and finite, (eager_same, n_graphs, w_sum, finite)
