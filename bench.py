#!/usr/bin/env python3
"""Headline benchmark: train-step samples/sec of the full joint pipeline (BASELINE.json configs[1]:
OS_CNN + Simplified_NF_WaveGlow + C_DAN (+CPC, GradNorm), synthetic univariate L=512, batch 256 per GPU).

    python bench.py --gpus N --steps K --warmup W
    N>1 either way: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` (the ranks
    read RANK / LOCAL_RANK / WORLD_SIZE), or as the plain command — bench.py then starts that launcher itself as a CHILD
    process before touching the GPU, relays rank 0's JSON line and exits with the child's code.
    Other BASELINE configs on one GPU: --sources 4 (configs[2]: four independent source->target pipelines, aggregate
    pairs/s), --c-in 9 --length 5000 --batch B (configs[3]), --length 1024 (configs[4]'s per-GPU shape).

A "step" is one pass of train_and_test.py:539-766 over one batch of 256 (target, source) pairs per GPU:
forward of every module, the GradNorm partial backward passes, one full backward, all optimiser updates.
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
# split-bf16 kernels: v_mfma_f32_32x32x16_bf16 = 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz = 2516.6 TFLOP/s dense (the
# "~2.5 PF" of the guide), and every algorithmic product costs three MFMAs (hi*hi + hi*lo + lo*hi)
BF16X3_MFMA_PEAK_TFLOPS = 2516.6 / 3.0
HBM_PEAK_GBS = 8000.0


def synthetic_batch(B, C_in, L, n_class, device, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C_in, L, generator=g)
    x = (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True)             # z-normalised per series (UCR convention)
    y = torch.randint(n_class, (B,), generator=g)
    return x.to(device), y.to(device)


def host_cores():
    """(threads to use, description): min(CPUs in this process's affinity mask, physical cores, cgroup CPU quota)."""
    affinity = len(os.sched_getaffinity(0))
    physical = None
    try:
        ids, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        ids.add((phys, core))
                    phys = core = None
        physical = len(ids) or None
    except OSError:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    n = min(v for v in (affinity, physical, quota) if v)
    return n, {"cpus_in_affinity": affinity, "physical_cores": physical, "cgroup_cpu_quota": quota, "host_logical_cpus": os.cpu_count()}


def _time_oracle_steps(step, steps: int, warm: bool = True) -> float:
    if warm:
        step()                                                                # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    return (time.perf_counter() - t0) / steps


def cpu_baseline(L: int, pairs: int, steps: int, c_in: int = 1, threads: int = 0, small_pairs: int = 16):
    """The CPU oracle (oracle/restatement.py, a port of the reference's step) timed on this host's cores.  Baseline only —
    never the product path.  Three measurements, all on `threads` threads = min(affinity, physical cores, cgroup quota)
    unless --cpu-threads says otherwise:
      * the headline: the joint step S2 at the METRIC's batch (`pairs`, default 256) — 1 warm-up + `steps` timed steps;
      * S1, the classifier-only step (train_and_test.py:153-171), same batch;
      * the small sample earlier rounds quoted (16 pairs/step, 3 steps), once more with autograd anomaly mode ON as the
        reference ships (train_and_test.py:24) — kept for continuity (BatchNorm / CPC couple the batch and per-op overheads
        weigh more at 16 pairs, so it is not the same workload)."""
    from oracle import restatement as R
    auto, host = host_cores()
    torch.set_num_threads(threads or auto)
    mk = lambda g, n: ((lambda x: (x - x.mean(-1, keepdim=True)) / x.std(-1, keepdim=True))(torch.randn(n, c_in, L, generator=g)),
                       torch.randint(4, (n,), generator=g))

    def joint(n_pairs, n_steps, warm=True):
        torch.manual_seed(1234)
        js = R.build_joint_step(L, c_in, L, c_in, 4, 4, seed=1234)
        g = torch.Generator().manual_seed(99)
        (x_t, y_t), (x_s, y_s) = mk(g, n_pairs), mk(g, n_pairs)
        return js, (x_t, y_t, x_s, y_s), _time_oracle_steps(lambda: js.step(x_t, y_t, x_s, y_s, epoch=0), n_steps, warm)
    # the 16-pair sample runs first: it is also the warm-up (threads, allocator, operator dispatch) of the full-batch step, which is
    # then timed WITHOUT a full-batch warm-up of its own — one more 80-150 s step would be most of the bench's budget
    small = None
    if small_pairs and small_pairs != pairs:
        js, batch, dts = joint(small_pairs, 3)
        small = {"pairs_per_step": small_pairs, "value": small_pairs / dts, "unit": "samples/s", "steps": 3}
        t0 = time.perf_counter()
        try:
            with torch.autograd.set_detect_anomaly(True, check_nan=True):
                js.step(*batch, epoch=0)
            small["anomaly_mode_on"] = {"value": small_pairs / (time.perf_counter() - t0), "unit": "samples/s", "steps": 1}
        except RuntimeError as e:                                             # anomaly mode raises on a NaN (degenerate toy sizes)
            small["anomaly_mode_on"] = {"error": str(e)[:200]}
        del js, batch
    _, _, dt = joint(pairs, steps, warm=small is None)
    out = {"value": pairs / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port", **host,
           "sample": f"joint step S2 (L={L}, C_in={c_in}) at the metric's batch: {pairs} pairs/step, {steps} timed step(s) "
                     f"({'warmed up by the small sample only' if small is not None else 'after 1 warm-up'}), {dt:.1f} s/step, "
                     f"autograd anomaly mode off, {torch.get_num_threads()} threads"}
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "?")
    except OSError:
        out["cpu_model"] = "?"
    # S1 at the same batch
    g = torch.Generator().manual_seed(7)
    fe_spec, clf_spec = R.train_specs(L, c_in)
    s1 = R.ClassifierStep(R.init_feature_extractor(fe_spec, g), R.init_classifier(clf_spec, 4, g), fe_spec, clf_spec)
    x, y = mk(g, pairs)
    dt1 = _time_oracle_steps(lambda: s1.step(x, y), 2)
    out["s1_classifier_step"] = {"value": pairs / dt1, "unit": "samples/s", "s_per_step": dt1, "batch": pairs, "steps": 2}
    if small is not None:
        out["small_sample"] = small
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher environment: start N ranks through torch.distributed.run as a child
    process (never exec: nothing here has touched the GPU yet, but a child keeps that true whatever is added later),
    pass its output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                          # dmabuf IPC only on this pool (RCCL needs it)
    print("[bench] launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def f32_mode_rate(args):
    """The same step with every GEMM on the exact-f32 MFMA (FST_MATH=f32), measured by a child process (the arithmetic is
    fixed at import).  Sits beside the split-bf16 headline so the narrower product arithmetic is visible in the line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(2, min(args.steps, 4))), "--warmup", "1",
           "--batch", str(args.batch), "--length", str(args.length), "--c-in", str(args.c_in), "--sources", str(args.sources),
           "--plain"]
    out = subprocess.run(cmd, env=dict(os.environ, FST_MATH="f32"), cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": (out.stderr or out.stdout)[-400:]}
    d = json.loads(lines[-1])
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "dtype": d["dtype"],
            "note": "same workload, FST_MATH=f32: v_mfma_f32_32x32x2_f32 everywhere (bit-exact fp32 FMA chains)"}


def classifier_step_rate(fst, device, B: int, L: int, steps: int = 30, warmup: int = 5):
    """S1 of SURVEY §8d: the classifier-only step (FE -> classifier -> CE -> backward -> RMSprop,
    train_and_test.py:153-171) on the same synthetic batch shape; eager launches, HIP-event timed."""
    tr = fst.ClassifierTrainer(L, 1, 4, device)
    x, y = synthetic_batch(B, 1, L, 4, device, 3000)
    for _ in range(warmup):
        tr.step(x, y)
    mode = "graph"
    try:
        tr.capture(x, y)
        run = lambda: tr.replay(x, y)
    except Exception as e:                                                    # noqa: BLE001 — report, then time eagerly
        print(f"[bench] S1 capture failed ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr, flush=True)
        mode, run = "eager", (lambda: tr.step(x, y))
        torch.cuda.synchronize()
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {"value": 1e3 * B / ms, "unit": "samples/s", "ms_per_step": ms, "steps": steps, "mode": mode,
            "workload": f"S1 classifier-only step (OS_CNN_res + OS_CNN + CE + RMSprop), univariate L={L}, batch {B}"}


def _timed(fn, reps: int = 10, warm: int = 2) -> float:
    """ms per call, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def north_star_extras(fst, ops, trainer, x_t, B: int, L: int):
    """The three figures BASELINE.json's north_star asks for besides the step rate: the omni-scale Conv1d
    sweep as algorithmic HBM GB/s (input once, features once) next to its FLOP rate, the CPC cross-Gram ("Gram GEMM")
    and the CDAN random-layer GEMM against their ceilings.  Forward kernels only, HIP-event timed, no autograd."""
    out = {}
    with torch.no_grad():
        fe = trainer.m["fe_t"]
        C_in, C = x_t.size(1), 50
        ms = _timed(lambda: fe(x_t))
        macs = 964 * C_in + 216900 + 16875 + 50 * C_in                       # live MACs per timestep
        out["omni_scale_fe_forward"] = {
            "ms": ms, "algorithmic_GBps": 4.0 * (C_in + C) * B * L / (ms * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS,
            "TFLOPps": 2.0 * macs * B * L / (ms * 1e-3) / 1e12,
            "counters": "profiles/r04_north_star_micro_hbm_traffic.csv, profiles/r04_north_star_micro_mfma_busy.csv (rocprofv3 --pmc passes on tools/north_star_micro.py)",
            "note": "OS_CNN_res forward incl. train-mode BatchNorm passes; layer 1 (216 900 live MACs/timestep) runs on "
                    "conv_win_rows_kernel / conv_win_bf3_kernel — see roofline.kernels for its own rate; the block is compute-bound, the GB/s "
                    "figure is input + features once"}
        feat = torch.randn(B, C, L, device=x_t.device)
        T = L // 2
        pred = torch.randn(T, B, C, device=x_t.device) * 0.3
        ms = _timed(lambda: ops.CPCNceFn.apply(feat, pred, 7, T))
        out["cpc_cross_gram"] = {"ms": ms, "counters": "profiles/r04_north_star_micro_hbm_traffic.csv, profiles/r04_north_star_micro_mfma_busy.csv", "TFLOPps": 2.0 * T * B * B * C / (ms * 1e-3) / 1e12,
                                 "peak_TFLOPps": BF16X3_MFMA_PEAK_TFLOPS if ops.MATH == "bf16x3" else F32_MFMA_PEAK_TFLOPS,
                                 "note": ("all T cross-Grams enc_i·pred_iT (K = C = 50 padded to 64): a transposing gather of the encodings + one "
                                          "workgroup per step on v_mfma_f32_32x32x16_bf16 with split operands (hi*hi + hi*lo + lo*hi), log-softmax + "
                                          "diagonal on the accumulators of the transposed tile; ms = both launches + the 256-slot loss sum; priced "
                                          "against the split-bf16 MFMA peak; counters: 54 MB moved for 39 MB of operands, MFMA-busy 0.09")
                                 if ops.MATH == "bf16x3" else
                                 "all T cross-Grams on v_mfma_f32_32x32x2_f32 (exact fp32); priced against the f32 MFMA peak"}
        rl = trainer.random_layer
        xf = torch.randn(B, C * L, device=x_t.device)
        R0, R1 = rl.random_matrix[0], rl.random_matrix[1]
        R0t = rl._rt(0)
        pr = torch.softmax(torch.randn(B, R1.size(0), device=x_t.device), 1)
        scale = 1.0 / R0.size(1) ** 0.5
        ms = _timed(lambda: ops.nt_gemm(xf, R0t, (pr, R1, scale)))
        ms_b = _timed(lambda: ops.nt_gemm(pr.new_empty(B, R0.size(1)).normal_(), R0))
        out["cdan_random_layer_gemm"] = {
            "ms": ms, "TFLOPps": 2.0 * B * C * L * R0.size(1) / (ms * 1e-3) / 1e12,
            "algorithmic_GBps": 4.0 * (R0.numel() + xf.numel() + B * R0.size(1)) / (ms * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBS,
            "matrix_GBps": 4.0 * R0.numel() / (ms * 1e-3) / 1e9, "data_gradient_ms": ms_b,
            "counters": "profiles/r04_north_star_micro_hbm_traffic.csv, profiles/r04_north_star_micro_mfma_busy.csv",
            "note": "RandomLayer.forward as ONE GEMM: 256 x 25600 x 1024 on the time-as-k kernel (fst_nt_gemm: split-bf16 MFMA, both "
                    "operands row-major with K contiguous, the 105 MB fixed matrix read once by the LDS-DMA ring, K split into 32 "
                    "partial slabs added in a fixed order - no atomics) with the class-side product, the 1/sqrt(1024) scale and the "
                    "Hadamard product in the reduce epilogue; data_gradient_ms: dy x R0 (256 x 1024 x 25600, no K split)"}
    return out


def kernel_peak(key: str) -> float:
    """Dense MFMA peak (algorithmic TFLOP/s) of the instruction a conv-engine kernel is built on."""
    bf3 = "bf3" in key or key.startswith(("wn_layer_", "wn_stack_")) or (key.startswith("conv_wgrad_kernel") and key.rstrip(">").endswith("true"))
    return BF16X3_MFMA_PEAK_TFLOPS if bf3 else F32_MFMA_PEAK_TFLOPS


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--length", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs per step of the CPU baseline (default: --batch, the metric's batch)")
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: min(affinity, physical cores, cgroup quota))")
    ap.add_argument("--c-in", type=int, default=1, help="input channels of both domains (configs[3]: 9)")
    ap.add_argument("--sources", type=int, default=1,
                    help="independent source->target pipelines trained side by side on this GPU (configs[2]: 4); "
                         "value = aggregate pairs/s")
    ap.add_argument("--plain", action="store_true", help="step rate only: no S1 / north-star extras / CPU / f32 legs")
    ap.add_argument("--mode", choices=["graph", "eager", "auto"], default="graph",
                    help="graph: the whole step is one captured hipGraph replayed per step — a failed capture is an error "
                         "(default); eager: launch by launch; auto: graph, falling back to eager with a note on stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # Rehearsal on a one-GPU box only (never set by the driver): FST_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # FST_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so the N>1 code path — two captured
    # graphs with the eager all-reduce between them, barrier + MAX timing — can be exercised without a multi-GPU node.
    if os.environ.get("FST_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import __graft_entry__ as entry
    if rank == 0:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):                          # stdout carries the one JSON line only
            entry.build()
    import torch.distributed as dist
    bucket = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FST_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()
    import feature_level_style_transfer_for_tsc_amd as fst
    from feature_level_style_transfer_for_tsc_amd import ops
    if world > 1:
        bucket = fst.GradBucket()

    torch.manual_seed(1234)                                                   # identical replicas on every rank
    cfg = fst.JointConfig(L_t=args.length, C_in_t=args.c_in, L_s=args.length, C_in_s=args.c_in, n_class_t=4, n_class_s=4)
    # --sources K: K independent pipelines (one per source domain, main.py:7-11), each with its own modules, optimisers
    # and source batch, sharing the target batch; one step = every pipeline stepped once.
    trainers = [fst.JointTrainer(cfg, device, bucket) for _ in range(args.sources)]
    trainer = trainers[0]
    x_t, y_t = synthetic_batch(args.batch, args.c_in, args.length, 4, device, 1000 + rank)
    src = [synthetic_batch(args.batch, args.c_in, args.length, 4, device, 2000 + rank + 97 * k) for k in range(args.sources)]
    x_s, y_s = src[0]
    T_half = (args.length // 2) // 2
    torch.manual_seed(4321)                                                   # CPC start indices: same on every rank

    def draw_t():
        return (int(torch.randint(T_half, (1,))), int(torch.randint(T_half, (1,))))

    mode = "graph" if args.mode == "auto" else args.mode
    if mode == "graph":
        try:
            for tr, (xs, ys) in zip(trainers, src):
                tr.capture(x_t, y_t, xs, ys, epoch=0)
        except Exception as e:                                                # noqa: BLE001
            if args.mode != "auto":
                raise SystemExit(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); --mode graph does not fall back "
                                 "(use --mode auto or --mode eager)")
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager", file=sys.stderr, flush=True)
            mode = "eager"
            torch.cuda.synchronize()

    def one_step():
        rep = None
        for tr, (xs, ys) in zip(trainers, src):
            if mode == "graph":
                rep = tr.replay(x_t, y_t, xs, ys, draw_t())
            else:
                rep = tr.step(x_t, y_t, xs, ys, epoch=0, t_samples=draw_t())
        return rep

    for i in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        rep = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rep = {k: v.clone() for k, v in rep.items()}
    # per-kernel durations for the roofline leg: HIP events around every conv-engine launch of ONE extra eager step
    # (same kernels, same shapes as the timed steps; a captured graph cannot carry timing events)
    timer = ops.KernelTimer()
    n_timer_steps = 1
    # every rank runs the step (it contains the gradient all-reduce); only rank 0 records the kernel timings
    ops.KERNEL_TIMER = timer if rank == 0 else None
    trainer.step(x_t, y_t, x_s, y_s, epoch=0, t_samples=draw_t())
    torch.cuda.synchronize()
    ops.KERNEL_TIMER = None
    ms_per_step = 1e3 * dt / args.steps
    value = world * args.sources * args.batch * args.steps / dt

    if rank == 0:
        ks = timer.summary()
        dom_key = max(ks, key=lambda k: ks[k]["total_ms"])
        dom = ks[dom_key]
        achieved = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        conv_ms = sum(v["total_ms"] for v in ks.values()) / n_timer_steps
        traffic, traffic_src = None, None
        # PMC passes cannot run inside bench.py: the newest committed counter summary that covers this kernel
        for tname in ("traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj["hbm_bytes_per_launch"].get(dom_key) is not None:
                    traffic, traffic_src = tj["hbm_bytes_per_launch"][dom_key], tj["source"]
                    break
        peak = kernel_peak(dom_key)
        mfma = {"achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_basis": ("bf16 dense MFMA peak 2516.6 TFLOP/s / 3 MFMAs per split-bf16 product" if peak != F32_MFMA_PEAK_TFLOPS
                               else "f32 MFMA peak")}
        gbs = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9
        hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"]}
        near = hbm if hbm["frac"] >= mfma["frac"] else mfma      # the roofline the dominant kernel is closer to
        roofline = {"bound": "hbm" if near is hbm else "mfma", "kernel": dom_key, "achieved": near["achieved"],
                    "peak": near["peak"], "unit": near["unit"], "frac": near["frac"], "mfma": mfma, "hbm": hbm,
                    "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                    "algorithmic_flop_per_launch": dom["flops"] / dom["launches"],
                    "avg_launch_us": dom["avg_us"], "launches_per_step": dom["launches"] / n_timer_steps,
                    "conv_engine_ms_per_step": conv_ms,
                    "kernels": {k: {"avg_us": round(v["avg_us"], 1), "launches_per_step": v["launches"] / n_timer_steps,
                                    "tflops": round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 2),
                                    "frac_of_mfma_peak": round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12 / kernel_peak(k), 3)}
                                for k, v in ks.items()}}
        metric = "train-step samples/sec (univariate TS, len=512, batch=256) at 1/2/4/8 MI355X"
        try:                                                                   # BASELINE.json's own string when present
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"] or metric
        except (OSError, KeyError, ValueError):
            pass
        line = {"metric": metric, "value": value, "unit": "samples/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16x3" if ops.MATH == "bf16x3" else "f32", "data": "synthetic",
                "config": {"workload": "%s: full joint step (OS_CNN_res x2 + OS_CNN x3 + WaveGlow(3,50,120) fwd x2 + infer "
                                       "+ CPC x2 + CDAN + GradNorm + RMSprop/Adam), %s L=%d, %d pairs/GPU%s"
                                       % (("configs[2]" if args.sources > 1 else "configs[3]" if args.c_in > 1 else
                                           "configs[4] per-GPU shape" if args.length == 1024 else "configs[1]"),
                                          "univariate" if args.c_in == 1 else f"{args.c_in}-channel", args.length, args.batch,
                                          f", {args.sources} independent source->target pipelines per step" if args.sources > 1 else ""),
                           "global_batch": world * args.batch, "seq_len": args.length, "c_in": args.c_in, "sources": args.sources,
                           "parallelism": f"dp{world}"},
                "mode": mode, "dist_backend": (os.environ.get("FST_BENCH_BACKEND", "nccl") if world > 1 else None),
                "arithmetic": ("fp32 storage; every GEMM-shaped product (WaveGlow convs, omni-scale window kernel, weight "
                               "gradients, random-layer GEMM) as hi*hi + hi*lo + lo*hi of round-to-nearest bf16 halves on "
                               "v_mfma_f32_32x32x16_bf16 with fp32 accumulation (error ~5e-6 of the output scale); exact-f32 "
                               "MFMA for omni-scale layers with < 8 input channels, shapes with L % 4 != 0 and the CPC "
                               "cross-Gram; everything pointwise in f32") if ops.MATH == "bf16x3" else
                              "f32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains) for every GEMM-shaped product",
                "losses": {k: float(rep[k]) for k in ("nf_t", "nf_s", "ce_t", "ce_s", "sl_t", "cdan")},
                "roofline": roofline}
        default_shape = args.c_in == 1 and args.sources == 1
        if world == 1 and not args.plain and default_shape:
            line["s1_classifier_step"] = classifier_step_rate(fst, device, args.batch, args.length)
            line["north_star_extras"] = north_star_extras(fst, ops, trainer, x_t, args.batch, args.length)
        if world == 1 and not args.plain and ops.MATH == "bf16x3":
            del trainers, trainer                                               # the child needs the HBM
            torch.cuda.empty_cache()
            line["f32_mode"] = f32_mode_rate(args)
        if world == 1 and not args.plain and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.length, args.cpu_pairs or args.batch, args.cpu_steps, args.c_in, args.cpu_threads)
            line["gpu_over_cpu"] = value / args.sources / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
