#!/usr/bin/env python3
"""Host-side sanitizer driver (run by tests/test_host_sanitizer_cpu.py under LD_PRELOAD=libclang_rt.asan).

The library given in FST_HIP_LIB is the HOST half of csrc/*.hip built with -fsanitize=address,undefined (device code objects
replaced by empty stand-ins: nothing can be launched, there is no GPU in the build container).  Every entry point is driven up
to its launch: argument checks, plan walking, extent arithmetic, pointer-table handling all run under ASan/UBSan; a launch
itself returns "no ROCm-capable device", which is the expected outcome for well-formed arguments.
Prints one line per call group; the test asserts a zero exit code and a clean sanitizer log."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from feature_level_style_transfer_for_tsc_amd import _lib                      # noqa: E402
from feature_level_style_transfer_for_tsc_amd.plan import Segment, build_plan  # noqa: E402
from feature_level_style_transfer_for_tsc_amd._lib import WSrc                 # noqa: E402

lib = _lib.load()
NO_DEVICE = 100          # hipErrorNoDevice: what a launch returns here
seen = {"bad": 0, "launch": 0, "value": 0}


def buf(n, dtype=np.float32):
    """16-byte aligned stand-in for a device buffer (never dereferenced: nothing launches)."""
    raw = np.zeros(n + 8, dtype=dtype)
    off = (-raw.ctypes.data // raw.itemsize) % (16 // raw.itemsize)
    a = raw[off: off + n]
    assert a.ctypes.data % 16 == 0
    return a


def P(a):
    return None if a is None else a.ctypes.data


def expect(rc, kind, what):
    if kind == "bad":
        assert rc == -1, f"{what}: expected an argument error, got rc={rc} ({lib.fst_last_error()})"
        assert lib.fst_last_error(), what
    elif kind == "launch":
        assert rc == NO_DEVICE, f"{what}: expected the launch to be reached (rc={NO_DEVICE}), got rc={rc} ({lib.fst_last_error()})"
    seen[kind] += 1


# ---- 1. every entry point with all-null / all-zero arguments: an error return or a plain value, never a crash
for name, (res, args) in _lib._SIGNATURES.items():
    if name == "fst_last_error":
        continue
    vals = []
    for t in args:
        if t in (ctypes.c_int, ctypes.c_int64, ctypes.c_int32):
            vals.append(0)
        elif t is ctypes.c_float:
            vals.append(0.0)
        else:
            vals.append(None)
    rc = getattr(lib, name)(*vals)
    assert rc != NO_DEVICE, f"{name} launched a kernel on all-null arguments"
    seen["value" if name.endswith(("_bytes", "_floats", "_ok", "_fits", "_slots", "fst_version")) else "bad"] += 1
print(f"null-argument calls: {len(_lib._SIGNATURES) - 1} entry points")

# ---- 2. conv engine: real plans walked on the host
B, L = 3, 96
for (M, C0, ntaps, dil, C1) in ((50, 50, 1, 1, 0), (240, 120, 3, 8, 25), (33, 7, 5, 1, 0)):
    segs = [Segment(0, C0, 0, ntaps)] + ([Segment(1, C1, ntaps // 2, ntaps // 2 + 1)] if C1 else [])
    plan = build_plan(M, segs, ntaps, dil, (ntaps - 1) * dil // 2)
    tab = np.ascontiguousarray(plan.table, dtype=np.int32)
    a_pk = buf(plan.packed_floats)
    w0, w1 = buf(M * C0 * ntaps), buf(max(1, M * C1))
    s0 = WSrc(P(w0), 0, C0 * ntaps, ntaps, 1)
    s1 = WSrc(P(w1), 0, C1, 1, 0)
    expect(lib.fst_pack_weights(P(tab), P(tab), tab.size, ctypes.byref(s0), ctypes.byref(s1) if C1 else None, M, 0, -1, 0, P(a_pk),
                                None), "launch", "fst_pack_weights")
    expect(lib.fst_pack_weights(P(tab), P(tab), tab.size - 3, ctypes.byref(s0), None, M, 0, -1, 0, P(a_pk), None), "bad",
           "fst_pack_weights (truncated plan)")
    x0, x1, y, bias = buf(B * C0 * L), buf(max(1, B * C1 * L)), buf(B * M * L), buf(M)
    expect(lib.fst_conv_gemm(P(x0), C0 * L, P(x1) if C1 else None, C1 * L, P(a_pk), P(tab), P(tab), tab.size, P(bias), P(y), M * L, None,
                             0, None, 0, M, M, B, L, M, 1, 1, 0, None), "launch", "fst_conv_gemm")
    expect(lib.fst_conv_gemm(P(x0), C0 * L - 1, None, 0, P(a_pk), P(tab), P(tab), tab.size, P(bias), P(y), M * L, None, 0, None, 0, M, M,
                             B, L, M, 1, 1, 0, None), "bad", "fst_conv_gemm (batch stride smaller than a sample)")
    bad = tab.copy(); bad[0] = 10 ** 6                                        # a chunk count the table cannot hold
    expect(lib.fst_conv_gemm(P(x0), C0 * L, None, 0, P(a_pk), P(bad), P(bad), bad.size, P(bias), P(y), M * L, None, 0, None, 0, M, M, B, L,
                             M, 1, 1, 0, None), "bad", "fst_conv_gemm (corrupt plan header)")
    dw, dw1 = buf(M * C0 * ntaps), buf(max(1, M * C1))
    expect(lib.fst_unpack_weights(P(tab), P(tab), tab.size, P(a_pk), M, P(dw), 0, C0 * ntaps, ntaps, 1, P(dw1) if C1 else None, 0, C1, 1,
                                  0, 1, None), "launch", "fst_unpack_weights")
    if C1:
        expect(lib.fst_unpack_weights(P(tab), P(tab), tab.size, P(a_pk), M, P(dw), 0, C0 * ntaps, ntaps, 1, None, 0, 0, 0, 0, 1, None),
               "bad", "fst_unpack_weights (second gradient target missing)")
print("conv engine: plans walked")

# ---- 3. pointer tables read on the host
for n_t in (1, 64, 70, 200):                                                   # (64 tensors per launch: chunking)
    ps = [buf(8) for _ in range(n_t)]
    arr = (ctypes.c_void_p * n_t)(*[P(p) for p in ps])
    ne = (ctypes.c_int64 * n_t)(*([8] * n_t))
    lr = np.full(n_t, 1e-3, dtype=np.float32)                                  # (a HOST array, one rate per tensor)
    expect(lib.fst_rmsprop_multi(arr, arr, arr, ne, P(lr), n_t, 0.99, 1e-8, None), "launch", f"fst_rmsprop_multi ({n_t} tensors)")
    step = buf(1)
    expect(lib.fst_adam_multi(arr, arr, arr, arr, ne, n_t, P(step), 1e-3, 0.9, 0.999, 1e-8, None), "launch", f"fst_adam_multi ({n_t} tensors)")
    ne[n_t - 1] = 0
    expect(lib.fst_rmsprop_multi(arr, arr, arr, ne, P(lr), n_t, 0.99, 1e-8, None), "bad" if n_t <= 64 else "value",
           "fst_rmsprop_multi (empty tensor)")
print("multi-tensor optimiser tables")

n, h, Bq, Lq = 120, 25, 4, 64
for n_sets in (1, 3):
    dg = [buf(Bq * 2 * n * Lq) for _ in range(n_sets)]
    a = [buf(Bq * n * Lq) for _ in range(n_sets)]
    u0 = [buf(Bq * h * Lq) for _ in range(n_sets)]
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[P(t) for t in ts])
    ws_n = lib.fst_wn_wgrad_workspace_floats(0, Bq, Lq, n, h, 0)
    assert ws_n > 0
    ws, dwi, dwc = buf(ws_n), buf(2 * n * n * 3), buf(2 * n * h)
    expect(lib.fst_wn_wgrad_in(arr(dg), arr(a), arr(u0), n_sets, h * Lq, P(dwi), P(dwc), P(ws), ws_n, Bq, Lq, n, h, 4, 0, Bq * n * Lq,
                               None), "launch", f"fst_wn_wgrad_in ({n_sets} sets)")
    expect(lib.fst_wn_wgrad_in(arr(dg), arr(a), arr(u0), n_sets, h * Lq, P(dwi), P(dwc), P(ws), 1000, Bq, Lq, n, h, 4, 0,
                               Bq * n * Lq, None), "bad", "fst_wn_wgrad_in (workspace too small)")
    expect(lib.fst_wn_wgrad_in(arr(dg), arr(a), arr(u0), n_sets, h * Lq, P(dwi), P(dwc), P(ws), ws_n, Bq, Lq, n, h, 2, 0, Bq * n * Lq,
                               None), "bad", "fst_wn_wgrad_in (dilation 2 without slack)")
    ts, d_a, d_out = [buf(Bq * 2 * n * Lq) for _ in range(n_sets)], a, [buf(Bq * n * Lq) for _ in range(n_sets)]
    ws_n = lib.fst_wn_wgrad_workspace_floats(1, Bq, Lq, n, 0, 0)
    ws, dwr = buf(ws_n), buf(2 * n * n)
    expect(lib.fst_wn_wgrad_rs(arr(d_a), arr(d_out), arr(ts), n_sets, P(dwr), P(ws), ws_n, 0, Bq, Lq, n, Bq * n * Lq, None), "launch",
           f"fst_wn_wgrad_rs ({n_sets} sets)")
expect(lib.fst_wn_wgrad_in(arr(dg), arr(a), arr(u0), 4, h * Lq, P(dwi), P(dwc), P(ws), ws_n, Bq, Lq, n, h, 4, 0, Bq * n * Lq, None), "bad",
       "fst_wn_wgrad_in (4 sets)")
print("time-as-k weight gradient: operand sets")

# ---- 4. fused WN layer launchers: image sizes and extents
img_b = lib.fst_wn_image_bytes(n, h)
assert img_b > 0
img = buf(img_b // 4)
a1, u1, ts1, out1, an1 = buf(Bq * n * Lq), buf(Bq * h * Lq), buf(Bq * 2 * n * Lq), buf(Bq * n * Lq), buf(Bq * n * Lq)
expect(lib.fst_wn_layer_fwd(P(a1), n * Lq, P(u1), h * Lq, P(img), img_b, P(ts1), None, P(an1), P(out1), 1, 0, Bq, Lq, n, h, 4,
                            Bq * n * Lq, None), "launch", "fst_wn_layer_fwd")
expect(lib.fst_wn_layer_fwd(P(a1), n * Lq, P(u1), h * Lq, P(img), img_b - 16, P(ts1), None, P(an1), P(out1), 1, 0, Bq, Lq, n, h, 4,
                            Bq * n * Lq, None), "bad", "fst_wn_layer_fwd (short image)")
expect(lib.fst_wn_layer_fwd(P(a1), n * Lq, P(u1), h * Lq, P(img), img_b, P(ts1), None, P(an1), P(out1), 1, 0, Bq, Lq, n, h, 4,
                            Bq * n * Lq + 1, None), "bad", "fst_wn_layer_fwd (element count)")
# the whole stack's backward: HOST tables of per-layer device pointers walked before the one launch
nl = 8
imgb = [buf(lib.fst_wn_bwd_image_bytes(n, int(i == nl - 1)) // 4) for i in range(nl)]
imgd = [buf(lib.fst_wn_dgrad_image_bytes(n) // 4) for _ in range(nl)]
tss = [buf(Bq * 2 * n * Lq) for _ in range(nl)]
dgs = [buf(Bq * 2 * n * Lq) for _ in range(nl)]
das = [buf(Bq * n * Lq) for _ in range(nl)]
rsb, rsd = [buf(256 * Bq) for _ in range(nl)], [buf(128 * Bq) for _ in range(nl)]
d_out1, du1 = buf(Bq * n * Lq), buf(Bq * h * Lq)
tab = lambda ts: (ctypes.c_void_p * len(ts))(*[P(t) for t in ts])
assert lib.fst_wn_stack_bwd_ok(n, h, Lq, nl) == 1 and lib.fst_wn_stack_bwd_ok(n, h, 1024, nl) == 0 and lib.fst_wn_stack_bwd_ok(n, h, Lq, 11) == 0
expect(lib.fst_wn_stack_bwd(tab(tss), tab(imgb), tab(imgd), tab(dgs), tab(das), tab(rsb), tab(rsd), P(d_out1), P(du1), h * Lq, nl, Bq, Lq,
                            n, h, Bq * n * Lq, None), "launch", "fst_wn_stack_bwd (full pass: every layer's dg / d_a kept)")
scratch = [das[0]] + [None] * (nl - 1)
expect(lib.fst_wn_stack_bwd(tab(tss), tab(imgb), tab(imgd), tab([dgs[0]] * nl), tab(scratch), None, None, P(d_out1), P(du1), h * Lq, nl,
                            Bq, Lq, n, h, Bq * n * Lq, None), "launch", "fst_wn_stack_bwd (partial pass: scratch dg, only layer 0's d_a)")
expect(lib.fst_wn_stack_bwd(tab(tss), tab(imgb), tab(imgd), tab(dgs), tab([None] * nl), None, None, P(d_out1), P(du1), h * Lq, nl, Bq, Lq,
                            n, h, Bq * n * Lq, None), "bad", "fst_wn_stack_bwd (layer 0's d_a missing)")
expect(lib.fst_wn_stack_bwd(tab(tss), tab(imgb), tab(imgd), tab(dgs), tab(das), tab(rsb), None, P(d_out1), P(du1), h * Lq, nl, Bq, Lq,
                            n, h, Bq * n * Lq, None), "bad", "fst_wn_stack_bwd (one row-sum table only)")
expect(lib.fst_wn_stack_bwd(tab(tss), tab(imgb), tab(imgd), tab(dgs), tab(das), None, None, P(d_out1), P(du1), h * Lq - 4, nl, Bq, Lq,
                            n, h, Bq * n * Lq, None), "bad", "fst_wn_stack_bwd (d_u0 batch stride smaller than a sample)")
assert lib.fst_cpc_nce_slots(256, 256, 50, 256) == 256 and lib.fst_cpc_workspace_floats(256, 256, 50, 256) == 256 * 256 * 50
assert lib.fst_cpc_workspace_floats(8, 16, 50, 512) == 8 * 16 * 2 * 3
print("fused WN layer launchers")

# ---- 5. NoiseTransfer / BatchNorm / row sums: shape checks
C, Ln = 50, 64
part = buf(2 * 8 * C * Ln)
expect(lib.fst_batch_sum(P(buf(16 * C * Ln)), P(buf(16 * C * Ln)), P(part), 16, C * Ln, 8, None), "launch", "fst_batch_sum")
expect(lib.fst_batch_sum(P(buf(16 * C * Ln)), None, P(part), 16, C * Ln + 2, 8, None), "bad", "fst_batch_sum (N % 4)")
expect(lib.fst_row_sum(P(buf(4 * 8 * 16)), 8 * 16, 4, 8, 16, P(buf(8)), None), "launch", "fst_row_sum")
expect(lib.fst_row_sum(P(buf(4 * 8 * 16)), 8 * 16 - 1, 4, 8, 16, P(buf(8)), None), "bad", "fst_row_sum (stride)")
print("pointwise launchers")

# ---- 6. fst_gemm: leading dimensions per layout, activation code, the K-split workspace
x, w, y = buf(256 * 1024), buf(1024 * 1024), buf(256 * 1024)
ws_n = lib.fst_gemm_workspace_floats(256, 1024, 1024)
assert ws_n == 0 or ws_n % (256 * 1024) == 0
ws = buf(max(ws_n, 4))
expect(lib.fst_gemm(P(x), 1024, 0, P(w), 1024, 0, P(y), 1024, 256, 1024, 1024, P(buf(1024)), 1, 0.0, P(ws), ws_n, None), "launch",
       "fst_gemm (y = x Wt + b, ReLU)")
expect(lib.fst_gemm(P(y), 1024, 0, P(w), 1024, 1, P(x), 1024, 256, 1024, 1024, None, 0, 0.0, P(ws), ws_n, None), "launch", "fst_gemm (dx = g W)")
expect(lib.fst_gemm(P(y), 1024, 1, P(x), 1024, 1, P(w), 1024, 1024, 1024, 256, None, 0, 0.0, P(ws), lib.fst_gemm_workspace_floats(1024, 1024, 256),
                    None), "launch", "fst_gemm (dW = gt x)")
expect(lib.fst_gemm(P(x), 50, 0, P(w), 50, 0, P(y), 1, 256, 1, 50, None, 2, 0.2, None, 0, None), "launch", "fst_gemm (one output column, K % 4 != 0)")
expect(lib.fst_gemm(P(x), 512, 0, P(w), 1024, 0, P(y), 1024, 256, 1024, 1024, None, 0, 0.0, P(ws), ws_n, None), "bad", "fst_gemm (lda < K)")
expect(lib.fst_gemm(P(y), 128, 1, P(x), 1024, 1, P(w), 1024, 1024, 1024, 256, None, 0, 0.0, P(ws), ws_n, None), "bad", "fst_gemm (k-major lda < M)")
expect(lib.fst_gemm(P(x), 1024, 0, P(w), 1024, 0, P(y), 512, 256, 1024, 1024, None, 0, 0.0, P(ws), ws_n, None), "bad", "fst_gemm (ldc < N)")
expect(lib.fst_gemm(P(x), 1024, 0, P(w), 1024, 0, P(y), 1024, 256, 1024, 1024, None, 3, 0.0, P(ws), ws_n, None), "bad", "fst_gemm (activation code)")
if ws_n > 0:
    expect(lib.fst_gemm(P(x), 1024, 0, P(w), 1024, 0, P(y), 1024, 256, 1024, 1024, None, 0, 0.0, P(ws), ws_n - 1, None), "bad", "fst_gemm (workspace)")
expect(lib.fst_act_bwd(P(x), P(y), P(buf(256 * 1024)), 256 * 1024, 0.2, None), "launch", "fst_act_bwd")
expect(lib.fst_act_bwd(P(x), None, P(y), 16, 0.0, None), "bad", "fst_act_bwd (null)")
print("dense products of the heads")
print(f"OK: {seen['bad']} argument errors returned, {seen['launch']} launches reached, {seen['value']} size queries")
