"""Autograd-visible ops of the hot path, each a thin host wrapper over the C ABI (include/fst_hip.h).

Everything here launches hand-written gfx950 kernels on torch's current HIP stream; torch supplies
device memory and the autograd graph only.  No op has a CPU implementation.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import dist as _dist
from ._lib import WSrc, check, ptr, stream_ptr
from .plan import Plan, Segment, build_plan, pick_mb

EPI_RELU, EPI_ACC2, EPI_ATOMIC, EPI_ACC1 = 1, 2, 4, 8
GEMM_BF16X3 = 16
WGRAD_SLABS = 32
# weight gradients: one partial-sum slab per K slice, added by the unpack pass (default) instead of fp32 atomics into one buffer
WGRAD_TWO_STAGE = os.environ.get("FST_WGRAD_ATOMICS", "0") != "1"
# Arithmetic of the pipelined forward / data-gradient GEMMs: "bf16x3" = split-bf16 operands on the bf16 matrix cores
# (hi*hi + hi*lo + lo*hi, fp32 accumulate, ~5e-6 of the output scale); "f32" = exact f32 MFMA everywhere.
MATH = os.environ.get("FST_MATH", "bf16x3")
LDS_BUDGET = 96 * 1024
LDS_MULTI_CHUNK = 44 * 1024
PIPE_C = 16            # channels per stage of the pipelined conv kernel (csrc/conv_engine.hip)

_PARTIAL_BACKWARD = False


class KernelTimer:
    """HIP-event timing of the conv-engine launches on the launch stream (used by bench.py's roofline leg).
    Keyed by the kernel template a launch dispatches to, so totals line up with rocprofv3's per-kernel stats."""

    def __init__(self, detail: bool = False):
        self.detail = detail       # True: key launches by shape as well (diagnostics)
        self.records = {}          # key -> list of (start_event, end_event, algorithmic_flops, algorithmic_hbm_bytes)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, key: str, start, flops: float, nbytes: float = 0.0):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.setdefault(key, []).append((start, ev, flops, nbytes))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            out[key] = {"launches": len(recs), "total_ms": ms, "avg_us": 1e3 * ms / len(recs),
                        "flops": sum(r[2] for r in recs), "bytes": sum(r[3] for r in recs)}
        return out


KERNEL_TIMER: Optional[KernelTimer] = None


def _vec16(plan: Plan, L: int, *tensors) -> bool:
    """Mirror of the host-side test in csrc/conv_engine.hip for the 16-byte-load kernel variants (timer keys only)."""
    shifts4 = plan.pad_left % 4 == 0 and (plan.ntaps == 1 or plan.dil % 4 == 0)
    ok = L % 4 == 0 and shifts4
    for t in tensors:
        if t is not None:
            ok = ok and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
    return ok


def bf3_ok(plan: Plan, L: int) -> bool:
    """Whether a conv_gemm launch of ``plan`` at sequence length L takes the split-bf16 path (activations of a
    multiple-of-4 length are 16-byte aligned by construction: channel stride L, batch stride C*L)."""
    if MATH != "bf16x3":
        return False
    return (plan.pipeable and L % 4 == 0) or plan.windowed16


def _plan_macs_per_step(plan: Plan, M: int) -> int:
    """Algorithmic MACs per (batch, timestep) of a plan: valid rows × live taps × real channels."""
    total = 0
    ch, en = plan.chunks(), plan.entries()
    rows_per_group = plan.MB * 32
    for g in range(plan.n_mgroups):
        rows = max(0, min(M, (g + 1) * rows_per_group) - g * rows_per_group)
        for q in range(plan.n_chunks):
            lo, hi = int(en[g, q, 0]), int(en[g, q, 1])
            if hi > lo:
                total += rows * (hi - lo) * int(ch[q, 2])
    return total


class partial_backward:
    """Context: a backward pass whose only wanted parameter gradients belong to convs flagged
    ``spec.always_weight_grad`` (GradNorm's shared OS_block); every other conv skips its weight-gradient kernels."""

    def __enter__(self):
        global _PARTIAL_BACKWARD
        self._prev, _PARTIAL_BACKWARD = _PARTIAL_BACKWARD, True
        return self

    def __exit__(self, *exc):
        global _PARTIAL_BACKWARD
        _PARTIAL_BACKWARD = self._prev
        return False


def _want_weight_grad(spec=None) -> bool:
    return (not _PARTIAL_BACKWARD) or bool(getattr(spec, "always_weight_grad", False))

Tensor = torch.Tensor


def _ncl(t: Tensor, name: str) -> Tuple[int, int]:
    """Check an activation view is [B, C, L] with unit time stride and channel stride L; return (bs, L)."""
    _lib.require_gpu_tensor(t, name)
    if t.dim() != 3 or t.stride(2) != 1 or (t.size(1) > 1 and t.stride(1) != t.size(2)):
        raise ValueError(f"{name}: expected NCL view with channel stride L, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0), t.size(2)


def _same_numel(*tensors: Optional[Tensor]) -> int:
    """Element count shared by the contiguous tensors a pointwise launch walks — taken from the tensors themselves, never
    from the (B, C, L) the launch is given, so the library can refuse a batch argument that does not describe them."""
    ts = [t for t in tensors if t is not None]
    n = ts[0].numel()
    for t in ts:
        if t.numel() != n or not t.is_contiguous():
            raise ValueError(f"pointwise launch over tensors of different extents / non-contiguous: {[tuple(x.shape) for x in ts]}")
    return n


def _gate_numel(acts: Tensor, *pairs: Tensor) -> int:
    """Element count of the [B, n, L] side of a gate launch; the [B, 2n, L] tensors must be exactly twice that."""
    n = _same_numel(acts)
    for t in pairs:
        if t.numel() != 2 * n or not t.is_contiguous():
            raise ValueError(f"gate launch: {tuple(t.shape)} is not the contiguous [B, 2n, L] partner of {tuple(acts.shape)}")
    return n


def _wsrc(t: Optional[Tensor], off0: int, sm: int, sc: int, st: int) -> WSrc:
    return WSrc(ptr(t) if t is not None else None, off0, sm, sc, st)


# --------------------------------------------------------------------------------------------------
# raw launches
# --------------------------------------------------------------------------------------------------
_PACK_CACHE: Optional[dict] = None


class pack_cache:
    """Context: within one train step the weights do not change, so each (plan, weight view) is packed once
    and reused by every forward / data-gradient call of the step (three WaveGlow passes, GradNorm's partial
    backward passes).  Entries keep their source tensors alive, so a data pointer cannot be recycled."""

    def __enter__(self):
        global _PACK_CACHE
        self._prev, _PACK_CACHE = _PACK_CACHE, {}
        return self

    def __exit__(self, *exc):
        global _PACK_CACHE
        _PACK_CACHE = self._prev
        return False


def pack_weights(plan: Plan, M: int, w0: Tensor, s0: Tuple[int, int, int, int], w1: Optional[Tensor] = None,
                 s1: Tuple[int, int, int, int] = (0, 0, 0, 0), parts=None, bf3: bool = False) -> Tensor:
    """Pack weights for ``plan``.  ``parts`` (optional) fills different M-group ranges from different weight
    tensors: a list of (g_begin, g_end, row_base, row_end, w, strides).  ``bf3``: the split-bf16 image."""
    if _PACK_CACHE is not None:
        key = (id(plan), bf3, M, w0.data_ptr(), w0._version, s0, None if w1 is None else (w1.data_ptr(), w1._version), s1,
               None if parts is None else tuple((p[0], p[1], p[2], p[3], p[4].data_ptr(), p[4]._version, p[5]) for p in parts))
        hit = _PACK_CACHE.get(key)
        if hit is not None:
            return hit[0]
        a = _pack_weights(plan, M, w0, s0, w1, s1, parts, bf3)
        _PACK_CACHE[key] = (a, w0, w1, plan, parts)
        return a
    return _pack_weights(plan, M, w0, s0, w1, s1, parts, bf3)


def _pack_weights(plan: Plan, M: int, w0: Tensor, s0, w1: Optional[Tensor], s1, parts=None, bf3: bool = False) -> Tensor:
    lib = _lib.load()
    a = torch.empty(plan.packed_floats_bf3 if bf3 else plan.packed_floats, device=w0.device, dtype=torch.float32)
    fn, who = (lib.fst_pack_weights_bf16x3, "fst_pack_weights_bf16x3") if bf3 else (lib.fst_pack_weights, "fst_pack_weights")
    second = (w1, s1)
    if parts is None:
        parts = [(0, -1, 0, M, w0, s0)]
    else:
        second = (None, (0, 0, 0, 0))
    for (g0, g1, row_base, row_end, w, sw) in parts:
        src0 = _wsrc(w, *sw)
        src1 = _wsrc(second[0], *second[1]) if second[0] is not None else None
        check(fn(ptr(plan.dev(w0.device)), plan.host_ptr(), plan.length, ctypes.byref(src0),
                 ctypes.byref(src1) if src1 is not None else None, row_end, g0, g1, row_base, ptr(a), stream_ptr()), who)
    return a


def unpack_weights(plan: Plan, M: int, a: Tensor, dw0: Tensor, s0, dw1: Optional[Tensor] = None, s1=(0, 0, 0, 0)) -> None:
    """``a``: a packed gradient, or [n_slabs, packed_floats] partial sums that are added on the way out."""
    lib = _lib.load()
    n_slabs = a.size(0) if a.dim() == 2 else 1
    check(lib.fst_unpack_weights(ptr(plan.dev(a.device)), plan.host_ptr(), plan.length, ptr(a), M, ptr(dw0), *s0,
                                 ptr(dw1), *s1, n_slabs, stream_ptr()), "fst_unpack_weights")


def conv_gemm(plan: Plan, a: Tensor, x0: Tensor, x1: Optional[Tensor], bias: Optional[Tensor], B: int, L: int, M: int,
              y: Optional[Tensor], res: Optional[Tensor] = None, y2: Optional[Tensor] = None, msplit: Optional[int] = None,
              nb: int = 1, ksplit: int = 1, flags: int = 0, m2_start: Optional[int] = None, bf3: bool = False) -> None:
    lib = _lib.load()
    if bf3:
        flags |= GEMM_BF16X3
    msplit = M if msplit is None else msplit
    m2_start = msplit if m2_start is None else m2_start
    x0_bs, _ = _ncl(x0, "x0")
    x1_bs = _ncl(x1, "x1")[0] if x1 is not None else 0
    y_bs = _ncl(y, "y")[0] if y is not None else 0
    res_bs = _ncl(res, "res")[0] if res is not None else 0
    y2_bs = _ncl(y2, "y2")[0] if y2 is not None else 0
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_conv_gemm(ptr(x0), x0_bs, ptr(x1), x1_bs, ptr(a), ptr(plan.dev(x0.device)), plan.host_ptr(),
                            plan.length, ptr(bias), ptr(y), y_bs, ptr(res), res_bs, ptr(y2), y2_bs, msplit, m2_start, B, L,
                            M, nb, ksplit, flags, stream_ptr()), "fst_conv_gemm")
    if t0 is not None:
        if bf3:
            key = f"conv_gemm_bf3_kernel<{plan.MB}, {nb}>" if plan.pipeable else f"conv_win_bf3_kernel<{plan.MB}, {nb}>"
        elif plan.pipeable and nb <= 2:
            key = f"conv_gemm_pipe_kernel<{plan.MB}, {nb}, {'true' if _vec16(plan, L, x0, x1) else 'false'}>"
        else:
            key = f"conv_gemm_kernel<{plan.MB}, {nb}>"
        if KERNEL_TIMER.detail:
            key += f" M={M} rec={plan.total_records} dil={plan.dil}"
        # algorithmic HBM bytes: every operand tensor once (inputs, residual / accumulate operands, outputs)
        rows_in = x0.size(1) + (x1.size(1) if x1 is not None else 0) + (res.size(1) if res is not None else 0)
        rows_out = (msplit if y is not None else 0) + ((M - m2_start) if y2 is not None else 0)
        if y2 is not None and flags & EPI_ACC2:
            rows_in += M - m2_start
        if flags & EPI_ACC1 and y is not None:
            rows_in += msplit
        KERNEL_TIMER.end(key, t0, 2.0 * B * L * _plan_macs_per_step(plan, M), 4.0 * B * L * (rows_in + rows_out))


def conv_wgrad(plan: Plan, x0: Tensor, x1: Optional[Tensor], dy: Tensor, dy2: Optional[Tensor], msplit: int, B: int,
               L: int, M: int, ksplit: int, x0_mul_off: int = 0) -> Tensor:
    lib = _lib.load()
    if WGRAD_TWO_STAGE:
        n_slabs = max(1, min(ksplit, B * ((L + 31) // 32)))                    # the library clamps the K split the same way
        da = torch.empty(n_slabs, plan.packed_floats, device=x0.device, dtype=torch.float32)
    else:
        da = torch.zeros(plan.packed_floats, device=x0.device, dtype=torch.float32)
    x0_bs, _ = _ncl(x0, "x0")
    x1_bs = _ncl(x1, "x1")[0] if x1 is not None else 0
    dy_bs, _ = _ncl(dy, "dy")
    dy2_bs = _ncl(dy2, "dy2")[0] if dy2 is not None else 0
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    bf3 = MATH == "bf16x3"
    check(lib.fst_conv_wgrad(ptr(x0), x0_bs, ptr(x1), x1_bs, ptr(dy), dy_bs, ptr(dy2), dy2_bs, msplit, ptr(da),
                             ptr(plan.dev(x0.device)), plan.host_ptr(), plan.length, B, L, M, ksplit,
                             (GEMM_BF16X3 if bf3 else 0) | (WGRAD_SLABS if WGRAD_TWO_STAGE else 0), x0_mul_off, stream_ptr()),
          "fst_conv_wgrad")
    if t0 is not None:
        wide = bool(((plan.entries()[:, :, 1] - plan.entries()[:, :, 0]) > 1).any())
        if wide:
            en = plan.entries()
            live = en[:, :, 1] > en[:, :, 0]
            starts4 = bool((((en[:, :, 0] * plan.dil - plan.pad_left) % 4 == 0) | ~live).all()) and \
                bool(((((en[:, :, 1] - 1 - en[:, :, 0]) * plan.dil) % 4 == 0) | ~live).all())
            vec = L % 4 == 0 and starts4 and all(t is None or (t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0)
                                                  for t in (x0, x1, dy, dy2))
        else:
            # single-tap windows: 1 = 16-byte staging, 2 = 16-byte staging that starts (shift mod 4) samples early
            aligned = L % 4 == 0 and all(t is None or (t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0) for t in (x0, x1, dy, dy2))
            vec = 0 if not aligned else (1 if _vec16(plan, L, x0, x1, dy, dy2) else 2)
        key = (f"conv_wgrad_kernel<{plan.MB // 4}, 32, {'true' if wide else 'false'}, {int(vec)}, "
               f"{'true' if bf3 else 'false'}>")
        if KERNEL_TIMER.detail:
            key += f" M={M} rec={plan.total_records} ksplit={ksplit}"
        rows_in = x0.size(1) + (x1.size(1) if x1 is not None else 0) + M
        KERNEL_TIMER.end(key, t0, 2.0 * B * L * _plan_macs_per_step(plan, M), 4.0 * B * L * rows_in + 8.0 * plan.packed_floats)
    return da


def row_sum(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    lib = _lib.load()
    bs, L = _ncl(x, "x")
    B, C = x.size(0), x.size(1)
    if out is None:
        out = torch.empty(C, device=x.device, dtype=torch.float32)          # written, not accumulated: no zero fill
    check(lib.fst_row_sum(ptr(x), bs, B, C, L, ptr(out), stream_ptr()), "fst_row_sum")
    return out


def wgrad_ksplit(B: int, L: int, n_wg_per_slice: int) -> int:
    """Split of (b, t) over workgroups.  All workgroups of the launch must be co-resident (2 per CU x 256 CUs): one
    more than that is a whole second round at the length of the first.  Workgroup id = slice + ksplit * item_group,
    and ids round-robin over the 8 XCDs, so with a multiple of 8 the item-groups that re-read one slice of dy sit on
    one XCD and share its L2.  Every slice adds an fp32-atomic pass over the packed gradient, so no more than fit."""
    tiles = B * ((L + 31) // 32)
    # a single item group (K <= 128 packed rows) is bound by its atomics — every slice adds a full pass over the
    # packed gradient — and measures best at 1.5 workgroups per CU (118 -> 93 us on the 120 -> 240 1x1 layers)
    target = 384 if n_wg_per_slice == 1 else 512
    k = max(1, target // max(1, n_wg_per_slice))
    if k >= 16:
        k -= k % 8
    return max(1, min(tiles, k))


# --------------------------------------------------------------------------------------------------
# ConvSpec: one convolution's shape + its forward / data-gradient / weight-gradient plans
# --------------------------------------------------------------------------------------------------
class ConvSpec:
    """y[b,m,t] = bias[m] + Σ_{c,k} w0[m,c,k]·x0[b,c,t+k·dil−pad_left] (+ Σ_c w1[m,c]·x1[b,c,t])."""

    def __init__(self, M: int, C0: int, ntaps: int = 1, dil: int = 1, pad_left: int = 0, C1: int = 0,
                 row_live: Optional[Sequence[Tuple[int, int]]] = None, dense_dw: bool = True):
        self.M, self.C0, self.ntaps, self.dil, self.pad_left, self.C1 = M, C0, ntaps, dil, pad_left, C1
        self.row_live = list(row_live) if row_live is not None else None
        self.dense_dw = dense_dw
        # a layer whose masked-tap gradients nobody reads (the classifier) still takes the dense many-tap kernel where that serves its
        # shape — it is faster than the live-tap item-table plan, and the masked taps then hold the reference's dense values, not zeros
        self.dense_if_fast = False
        if C1:
            if pad_left % dil or not (0 <= pad_left // dil < ntaps):
                raise ValueError("the 1x1 side input needs a tap with zero offset")
        self.x1_tap = pad_left // dil if C1 else 0
        self._plans: Dict[Tuple[str, int], Plan] = {}
        self.mb = 1 if self.row_live is not None and ntaps > 2 else pick_mb(M)

    def dense_tap_wgrad_ok(self, B: int, L: int, x0: Tensor, dy: Tensor) -> bool:
        """Whether ``dense_tap_wgrad`` serves this conv's weight gradient: a DENSE gradient (quirk Q1 layers, or no tap mask) over 5..96
        taps at dilation 1, no side input, at most 256 output rows, L % 32 == 0, split-bf16 mode, contiguous 16-byte aligned operands."""
        if self.C1 or MATH != "bf16x3" or not (self.dense_dw or self.row_live is None or self.dense_if_fast) or self.dil != 1:
            return False
        if os.environ.get("FST_DENSE_TAP_WGRAD", "1") == "0" or not (x0.is_contiguous() and dy.is_contiguous()):
            return False
        if x0.data_ptr() % 16 or dy.data_ptr() % 16 or dy.size(1) != self.M or x0.size(1) != self.C0:
            return False
        return bool(_lib.load().fst_dense_tap_wgrad_ok(B, L, self.M, self.C0, self.ntaps, self.pad_left))

    def tap_wgrad_ok(self, B: int, L: int, x0: Tensor, dy: Tensor) -> bool:
        """Whether ``tap_wgrad`` (csrc/wn_wgrad.hip) serves this conv's DENSE weight gradient: no side input, every tap wanted (a dense
        plan, or no tap mask at all), at most four taps, split-bf16 mode, contiguous operands — and slack around x when a tap is shifted
        by 1-3 samples.  Worth it only for layers the item-table kernel runs badly (many channels, few taps)."""
        if self.C1 or MATH != "bf16x3" or not (self.dense_dw or self.row_live is None) or self.ntaps < 2 or self.C0 < 64:
            return False
        # OFF by default: measured at the one layer it fits (225 → 50 channels, two taps, B=256, L=512) it takes 128 µs against the
        # item-table kernel's 85 — with 50 output rows the 256-row dy ring is four fifths padding.  FST_TAP_WGRAD=1 enables it.
        if os.environ.get("FST_TAP_WGRAD", "0") != "1" or not (x0.is_contiguous() and dy.is_contiguous()):
            return False
        if x0.data_ptr() % 16 or dy.data_ptr() % 16:
            return False
        served = _lib.load().fst_tap_wgrad_ok(B, L, self.M, self.C0, self.ntaps, self.dil, self.pad_left)
        return served == 1 or (served == 2 and has_slack(x0))

    # ---- heuristics
    def _halo(self) -> int:
        return (self.ntaps - 1) * self.dil

    def nb_for(self, B: int, L: int, mb: int, windowed_c: int, halo: int) -> int:
        """N-blocks (32 timesteps each) per wave.  windowed_c > 0: the whole-window kernel (omni-scale layers with
        many taps; channels are chunked to fit the LDS budget by ``_chunking``); otherwise the pipelined kernel,
        which has NB ∈ {1, 2}."""
        tiles128 = (L + 127) // 128
        win_bf3 = windowed_c >= 8 and MATH == "bf16x3"
        if win_bf3:
            windowed_c, mb = min(windowed_c, PIPE_C), min(mb, 2)               # bf16 window kernel: 16-channel slots, MB <= 2
        best = 1
        forced = int(os.environ.get("FST_WIN_NB", "0")) if windowed_c else 0           # diagnostics: force the window kernel's tile width
        if forced in (1, 2, 4) and mb * forced <= 8 and forced <= max(1, tiles128):
            return forced
        for nb in ((4, 2, 1) if windowed_c else (2, 1)):
            if mb * nb > (4 if win_bf3 else 8) or nb > max(1, tiles128):       # bf16 window kernel: 8 tiles per wave spill (not built)
                continue
            if windowed_c and windowed_c <= 64 and ((windowed_c + 1) & ~1) * (128 * nb + halo) * 4 > LDS_BUDGET:
                continue
            if windowed_c > 64 and nb > 2:
                continue
            best = nb
            if B * ((L + 128 * nb - 1) // (128 * nb)) >= 512:
                break
        return best

    def _fwd_segments(self) -> List[Segment]:
        segs = [Segment(0, self.C0, 0, self.ntaps)]
        if self.C1:
            segs.append(Segment(1, self.C1, self.x1_tap, self.x1_tap + 1))
        return segs

    def _windowed(self, channels: int) -> bool:
        """Many dense taps over few channels (omni-scale layers): stage one [C][T+halo] window and slide the taps
        over it.  Everything else is cut into single-tap 16-channel stages for the pipelined kernel."""
        return self.dil == 1 and self.ntaps > 3

    def _chunking(self, channels: int, nb: int, ntaps: int) -> Tuple[int, bool]:
        """(chunk_c, split_taps)"""
        if not self._windowed(channels):
            return PIPE_C, True
        if MATH == "bf16x3" and channels >= 8:
            return PIPE_C, False                                                # bf16 window kernel: 16-channel chunks, all taps
        halo = (ntaps - 1) * self.dil
        cap = (LDS_BUDGET // (4 * (128 * nb + halo))) & ~1
        if channels <= cap:
            return (channels + 1) & ~1, False
        cap = (LDS_MULTI_CHUNK // (4 * (128 * nb + halo))) & ~1                    # several chunks: keep 2+ workgroups per CU
        n = (channels + cap - 1) // cap
        return ((channels + n - 1) // n + 1) & ~1, False

    # ---- plans
    def fwd_plan(self, nb: int) -> Plan:
        key = ("fwd", nb)
        if key not in self._plans:
            chunk_c, split = self._chunking(max(self.C0, self.C1), nb, self.ntaps)
            mb = min(self.mb, 2) if (chunk_c == PIPE_C and not split) else self.mb     # bf16 window kernel: MB <= 2
            self._plans[key] = build_plan(self.M, self._fwd_segments(), self.ntaps, self.dil, self.pad_left, MB=mb,
                                          chunk_c=chunk_c, split_taps=split, row_live=self.row_live)
        return self._plans[key]

    def dx0_plan(self, nb: int) -> Plan:
        key = ("dx0", nb)
        if key not in self._plans:
            col_live = None
            if self.row_live is not None:
                col_live = [(self.ntaps - hi, self.ntaps - lo) for lo, hi in self.row_live]
            chunk_c, split = self._chunking(self.M, nb, self.ntaps)
            mb = min(pick_mb(self.C0), 2) if (chunk_c == PIPE_C and not split) else None   # bf16 window kernel: MB <= 2
            self._plans[key] = build_plan(self.C0, [Segment(0, self.M, 0, self.ntaps, col_live)], self.ntaps, self.dil,
                                          self._halo() - self.pad_left, MB=mb, chunk_c=chunk_c, split_taps=split)
        return self._plans[key]

    def dx1_plan(self, nb: int) -> Plan:
        key = ("dx1", nb)
        if key not in self._plans:
            self._plans[key] = build_plan(self.C1, [Segment(0, self.M, 0, 1)], 1, 1, 0, chunk_c=PIPE_C)
        return self._plans[key]

    def wg_plan(self) -> Plan:
        key = ("wg", 0)
        if key not in self._plans:
            cmax = max(self.C0, self.C1)
            if self._windowed(cmax):
                n = (cmax + 63) // 64                             # [<=64 channels][32+halo] windows, all taps
                chunk_c, split = ((cmax + n - 1) // n + 1) & ~1, False
            else:
                chunk_c, split = 32, True                         # single-tap 32-channel chunks = one row-block each
            self._plans[key] = build_plan(self.M, self._fwd_segments(), self.ntaps, self.dil, self.pad_left,
                                          MB=4 if self.M <= 128 else 8, chunk_c=chunk_c, split_taps=split,
                                          row_live=None if self.dense_dw else self.row_live, with_items=True)
        return self._plans[key]

    # ---- weight views
    def s_w0(self):
        return (0, self.C0 * self.ntaps, self.ntaps, 1)

    def s_w1(self):
        return (0, self.C1, 1, 0)

    def s_w0_T(self):
        """rows = input channel, K-channel = output row, taps flipped"""
        return (self.ntaps - 1, self.ntaps, self.C0 * self.ntaps, -1)

    def s_w1_T(self):
        return (0, 1, self.C1, 0)

    # ---- launches
    def forward(self, x0: Tensor, x1: Optional[Tensor], w0: Tensor, w1: Optional[Tensor], bias: Optional[Tensor],
                y: Optional[Tensor] = None, res: Optional[Tensor] = None, y2: Optional[Tensor] = None,
                msplit: Optional[int] = None, flags: int = 0) -> Tensor:
        B, L = x0.size(0), x0.size(2)
        windowed = self.C0 if self._windowed(max(self.C0, self.C1)) else 0
        nb = self.nb_for(B, L, self.mb, windowed, self._halo())
        plan = self.fwd_plan(nb)
        bf3 = bf3_ok(plan, L)
        a = pack_weights(plan, self.M, w0, self.s_w0(), w1, self.s_w1(), bf3=bf3)
        if y is None and (msplit is None or msplit > 0):
            y = torch.empty(B, self.M if msplit is None else msplit, L, device=x0.device, dtype=torch.float32)
        conv_gemm(plan, a, x0, x1, bias, B, L, self.M, y, res, y2, msplit, nb=nb, flags=flags, bf3=bf3)
        return y

    def grad_x0(self, dy: Tensor, w0: Tensor, out: Optional[Tensor] = None, res: Optional[Tensor] = None,
                flags: int = 0) -> Tensor:
        B, L = dy.size(0), dy.size(2)
        mb = pick_mb(self.C0)
        windowed = self.M if self._windowed(self.M) else 0
        nb = self.nb_for(B, L, mb, windowed, self._halo())
        plan = self.dx0_plan(nb)
        bf3 = bf3_ok(plan, L)
        a = pack_weights(plan, self.C0, w0, self.s_w0_T(), bf3=bf3)
        if out is None:
            out = torch.empty(B, self.C0, L, device=dy.device, dtype=torch.float32)
        conv_gemm(plan, a, dy, None, None, B, L, self.C0, out, res, nb=nb, flags=flags, bf3=bf3)
        return out

    def dx01_plan(self, nb: int) -> Plan:
        """Fused data gradient of both inputs: rows [0, C0) = d x0 (all taps, flipped); rows [R, R+C1) = d x1
        (the zero-offset tap only), R = C0 rounded up to an M-group — one pass over dy instead of two."""
        key = ("dx01", nb)
        if key not in self._plans:
            mb = pick_mb(self.C0)
            R = ((self.C0 + mb * 32 - 1) // (mb * 32)) * (mb * 32)
            tf = self.ntaps - 1 - self.x1_tap
            live = [(0, self.ntaps)] * R + [(tf, tf + 1)] * self.C1
            self._plans[key] = build_plan(R + self.C1, [Segment(0, self.M, 0, self.ntaps)], self.ntaps, self.dil,
                                          self._halo() - self.pad_left, MB=mb, chunk_c=PIPE_C, split_taps=True,
                                          row_live=live)
        return self._plans[key]

    def grad_x01(self, dy: Tensor, w0: Tensor, w1: Tensor, res0: Optional[Tensor], acc1: Tensor) -> Tensor:
        """returns d x0 = res0 + conv_T(dy, w0);  acc1 += conv_T(dy, w1)  — one launch."""
        B, L = dy.size(0), dy.size(2)
        mb = pick_mb(self.C0)
        nb = self.nb_for(B, L, mb, 0, 0)
        plan = self.dx01_plan(nb)
        R = plan.M - self.C1
        n_g0 = R // (mb * 32)
        bf3 = bf3_ok(plan, L)
        a = pack_weights(plan, plan.M, w0, self.s_w0_T(), parts=[(0, n_g0, 0, self.C0, w0, self.s_w0_T()),
                                                                 (n_g0, plan.n_mgroups, R, plan.M, w1, self.s_w1_T())],
                         bf3=bf3)
        out = torch.empty(B, self.C0, L, device=dy.device, dtype=torch.float32)
        conv_gemm(plan, a, dy, None, None, B, L, plan.M, out, res0, acc1, msplit=self.C0, nb=nb, flags=EPI_ACC2,
                  m2_start=R, bf3=bf3)
        return out

    def grad_x1(self, dy: Tensor, w1: Tensor, out: Optional[Tensor] = None, flags: int = 0) -> Tensor:
        B, L = dy.size(0), dy.size(2)
        nb = self.nb_for(B, L, pick_mb(self.C1), 0, 0)
        plan = self.dx1_plan(nb)
        bf3 = bf3_ok(plan, L)
        a = pack_weights(plan, self.C1, w1, self.s_w1_T(), bf3=bf3)
        if out is None:
            out = torch.empty(B, self.C1, L, device=dy.device, dtype=torch.float32)
        conv_gemm(plan, a, dy, None, None, B, L, self.C1, out, nb=nb, flags=flags, bf3=bf3)
        return out

    def grad_w(self, x0: Tensor, x1: Optional[Tensor], dy: Tensor, dy2: Optional[Tensor] = None,
               msplit: Optional[int] = None, x0_mul_off: int = 0, out1: Optional[Tensor] = None,
               out0: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """``x0_mul_off``: the x operand is x0[i]·x0[i + x0_mul_off] (x0 = a row slice of the saved gate halves).
        ``out0`` / ``out1``: contiguous [M, C0, ntaps] / [M, C1, 1] tensors (segments of a flat gradient, zero-filled when
        the plan does not write every element) that receive the gradients in place."""
        B, L = x0.size(0), x0.size(2)
        if x1 is None and dy2 is None and msplit is None and not x0_mul_off and self.dense_tap_wgrad_ok(B, L, x0, dy):
            # the dense gradient of an omni-scale layer (Q1): every tap from eight pre-shifted copies of one staged window per channel
            dw0 = out0 if out0 is not None else torch.empty(self.M, self.C0, self.ntaps, device=x0.device, dtype=torch.float32)
            dense_tap_wgrad(dy, x0, dw0, self.M, self.C0, self.ntaps, self.pad_left)
            return dw0, None
        if x1 is None and dy2 is None and msplit is None and not x0_mul_off and self.tap_wgrad_ok(B, L, x0, dy):
            # dense gradient of a conv with a few taps: the time-as-k kernel (each tap a k-row segment with its own shift)
            dw0 = out0 if out0 is not None else torch.empty(self.M, self.C0, self.ntaps, device=x0.device, dtype=torch.float32)
            tap_wgrad(dy, x0, dw0, self.M, self.C0, self.ntaps, self.dil, self.pad_left)
            return dw0, None
        plan = self.wg_plan()
        n_wg = max(1, len(plan.items()) // 4)
        da = conv_wgrad(plan, x0, x1, dy, dy2, self.M if msplit is None else msplit, B, L, self.M,
                        wgrad_ksplit(B, L, n_wg), x0_mul_off)
        # a dense plan (every tap of every row) makes unpack write every element: no zero fill needed
        alloc = torch.empty if (self.dense_dw or self.row_live is None) else torch.zeros
        if out0 is not None:
            assert out0.shape == (self.M, self.C0, self.ntaps) and out0.is_contiguous() and out0.dtype == torch.float32
            dw0 = out0
        else:
            dw0 = alloc(self.M, self.C0, self.ntaps, device=x0.device, dtype=torch.float32)
        dw1 = None
        if self.C1:
            if out1 is not None:
                assert out1.shape == (self.M, self.C1, 1) and out1.is_contiguous() and out1.dtype == torch.float32
                dw1 = out1                                # (a zero-filled target when the plan does not write every element)
            else:
                dw1 = alloc(self.M, self.C1, 1, device=x0.device, dtype=torch.float32)
        unpack_weights(plan, self.M, da, dw0, self.s_w0(), dw1, self.s_w1())
        return dw0, dw1


# --------------------------------------------------------------------------------------------------
# generic conv as an autograd op
# --------------------------------------------------------------------------------------------------
class ConvFn(torch.autograd.Function):
    """Conv1d (optionally dilated / omni-scale-masked) — F.conv1d call sites of the hot path."""

    @staticmethod
    def forward(ctx, spec: ConvSpec, x: Tensor, w: Tensor, bias: Optional[Tensor]):
        ctx.spec = spec
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return spec.forward(x, None, w, None, bias)

    @staticmethod
    def backward(ctx, dy):
        spec: ConvSpec = ctx.spec
        x, w = ctx.saved_tensors
        rs = _ROW_SUMS.take(dy)                                # Σ_t dy per (sample, channel) if the producer of dy left them
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dx = spec.grad_x0(dy, w)
        if ctx.needs_input_grad[2] and _want_weight_grad(spec):
            dw, _ = spec.grad_w(x, None, dy)
        if ctx.has_bias and ctx.needs_input_grad[3] and _want_weight_grad(spec):
            db = rs.sum(dim=0) if rs is not None else row_sum(dy)
        return None, dx, dw, db


def conv1d(spec: ConvSpec, x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    return ConvFn.apply(spec, x, w, bias)


def relu_bwd(dy: Tensor, y: Tensor) -> Tensor:
    """dy·[y > 0] — the backward of a ReLU that ran in a GEMM / conv epilogue (``y`` = its output)."""
    dy = dy.contiguous()
    out = torch.empty_like(dy)
    check(_lib.load().fst_relu_bwd(ptr(dy), ptr(y), ptr(out), _same_numel(dy, y), stream_ptr()), "fst_relu_bwd")
    return out


class ConvReluFn(torch.autograd.Function):
    """relu(conv1d(x, w) + bias), the ReLU in the GEMM epilogue (DimensionUnification's channel conv, widgets.py:76-77)."""

    @staticmethod
    def forward(ctx, spec: ConvSpec, x: Tensor, w: Tensor, bias: Optional[Tensor]):
        y = spec.forward(x, None, w, None, bias, flags=EPI_RELU)
        ctx.spec, ctx.has_bias = spec, bias is not None
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        spec: ConvSpec = ctx.spec
        x, w, y = ctx.saved_tensors
        g = relu_bwd(dy, y)
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dx = spec.grad_x0(g, w)
        if ctx.needs_input_grad[2] and _want_weight_grad(spec):
            dw, _ = spec.grad_w(x, None, g)
        if ctx.has_bias and ctx.needs_input_grad[3] and _want_weight_grad(spec):
            db = row_sum(g)
        return None, dx, dw, db


ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2       # FST_ACT_* of include/fst_hip.h


def gemm_ok(*ts: Tensor) -> bool:
    """Whether ``gemm`` serves these operands: split-bf16 mode, fp32 on the GPU (any shape, any row pitch)."""
    return MATH == "bf16x3" and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 for t in ts)


def gemm(A: Tensor, ta: bool, B: Tensor, tb: bool, bias: Optional[Tensor] = None, act: int = ACT_NONE, slope: float = 0.0) -> Tensor:
    """C[m, n] = act(Σ_k A(m,k)·B(n,k) + bias[n]) on the matrix cores in split-bf16 (fst_gemm).  ``A`` is a row-major matrix holding
    A(m,k) at [m, k] (``ta`` False) or at [k, m] (``ta`` True: the reduction index runs down the rows); ``B`` likewise with B(n,k) —
    Linear forward ``gemm(x, False, W, False, b)``, its data gradient ``gemm(g, False, W, True)``, its weight gradient
    ``gemm(g, True, x, True)``: no transposed copies.  K splits land in slabs added in a fixed order (deterministic)."""
    if not gemm_ok(A, B):
        raise ValueError(f"gemm: operands {tuple(A.shape)} / {tuple(B.shape)} must be fp32 matrices with unit column stride on the GPU "
                         f"(split-bf16 mode; FST_MATH={MATH})")
    lib = _lib.load()
    K, M = (A.shape if ta else A.shape[::-1])
    Kb, N = (B.shape if tb else B.shape[::-1])
    if K != Kb or (bias is not None and (bias.numel() != N or not bias.is_contiguous())):
        raise ValueError(f"gemm: A {tuple(A.shape)} (ta={ta}) and B {tuple(B.shape)} (tb={tb}) do not share a reduction length, or bias is not [N]")
    C = torch.empty(M, N, device=A.device, dtype=torch.float32)
    ws_n = lib.fst_gemm_workspace_floats(M, N, K)
    ws = torch.empty(ws_n, device=A.device, dtype=torch.float32) if ws_n > 0 else None
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_gemm(ptr(A), A.stride(0), int(ta), ptr(B), B.stride(0), int(tb), ptr(C), N, M, N, K, ptr(bias), act, float(slope),
                       ptr(ws), ws_n, stream_ptr()), "fst_gemm")
    if t0 is not None:
        KERNEL_TIMER.end("gemm_bf3_kernel bf3", t0, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N))
    return C


def act_bwd(dy: Tensor, y: Tensor, slope: float) -> Tensor:
    """dy·act'(y) from the activation's output: dy where y > 0, slope·dy elsewhere (ReLU: slope 0)."""
    dy = dy.contiguous()
    out = torch.empty_like(dy)
    check(_lib.load().fst_act_bwd(ptr(dy), ptr(y), ptr(out), _same_numel(dy, y), float(slope), stream_ptr()), "fst_act_bwd")
    return out


class LinearActFn(torch.autograd.Function):
    """act(x @ Wᵀ + b) over the last dimension — nn.Linear (+ nn.ReLU / nn.LeakyReLU) of the heads (DimensionUnification's length
    GEMM widgets.py:73-75, the CDAN discriminator :113-131, FeatureDiscriminatorforSource :32-42, the classifiers' last layer
    OS_CNN.py:268): bias and activation in the GEMM's epilogue, the three products of the layer on fst_gemm without transposed copies."""

    @staticmethod
    def forward(ctx, x: Tensor, W: Tensor, b: Optional[Tensor], act: int, slope: float):
        x2 = x.reshape(-1, x.size(-1))
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        y = gemm(x2, False, W, False, b, act, slope)
        ctx.save_for_backward(x2, W, y if act != ACT_NONE else None)
        ctx.lead, ctx.act, ctx.slope, ctx.has_bias = x.shape[:-1], act, slope, b is not None
        return y.view(*ctx.lead, W.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        g = dy.reshape(-1, W.size(0))
        g = act_bwd(g, y, ctx.slope if ctx.act == ACT_LEAKY else 0.0) if ctx.act != ACT_NONE else g.contiguous()
        dx = gemm(g, False, W, True).view(*ctx.lead, W.size(1)) if ctx.needs_input_grad[0] else None
        dW = gemm(g, True, x2, True) if ctx.needs_input_grad[1] and _want_weight_grad() else None
        db = g.sum(dim=0) if ctx.has_bias and ctx.needs_input_grad[2] and _want_weight_grad() else None
        return dx, dW, db, None, None


def linear_act(x: Tensor, lin: "torch.nn.Linear", act: int = ACT_NONE, slope: float = 0.0) -> Tensor:
    """``act(lin(x))``: on fst_gemm in split-bf16 mode; with FST_MATH=f32 the library's exact-f32 GEMM and the aten activation, as the
    reference runs them."""
    if x.is_cuda and x.dtype == torch.float32 and MATH == "bf16x3":
        return LinearActFn.apply(x, lin.weight, lin.bias, act, slope)
    h = torch.nn.functional.linear(x, lin.weight, lin.bias)
    if act == ACT_RELU:
        return torch.nn.functional.relu(h)
    if act == ACT_LEAKY:
        return torch.nn.functional.leaky_relu(h, slope)
    return h


def mask_taps_(w: Tensor, lo: Tensor, hi: Tensor) -> None:
    """W ← W ⊙ mask in place on ``w.data`` (OS_CNN.py:68 re-masks ``.data`` every forward)."""
    lib = _lib.load()
    M, C, K = w.shape
    check(lib.fst_mask_taps(ptr(w), ptr(lo), ptr(hi), M, C, K, stream_ptr()), "fst_mask_taps")


# --------------------------------------------------------------------------------------------------
# BatchNorm1d (+ReLU, + second BN'd branch) over (B, L)
# --------------------------------------------------------------------------------------------------
BN_SLOTS = 16      # FST_BN_SLOTS of include/fst_hip.h: workgroups (= partial results) per channel of the BatchNorm reductions


class _RowSums:
    """Per-(sample, channel) sums Σ_t dx that a backward launch has already formed, handed from the op that produced ``dx`` to the op
    that consumes it as its output cotangent (BatchNorm backward → the conv in front of it) — as an attribute of the tensor object,
    which autograd passes on unchanged when the producer is the only contributor to that cotangent.  ``take`` returns them only if
    the tensor is still the one they were computed from (same object, same version counter, same shape): an accumulated or
    otherwise rewritten cotangent falls back to the caller's own reduction."""

    @staticmethod
    def attach(dx: Tensor, rs: Tensor) -> None:
        dx._fst_row_sums = (rs, dx._version)

    @staticmethod
    def take(dy: Tensor) -> Optional[Tensor]:
        tag = getattr(dy, "_fst_row_sums", None)
        if tag is None or dy.dim() != 3:
            return None
        rs, version = tag
        if version != dy._version or tuple(rs.shape) != tuple(dy.shape[:2]) or not dy.is_contiguous():
            return None
        return rs


_ROW_SUMS = _RowSums()


def _bn_stats(y: Tensor, gamma: Tensor, beta: Tensor, rmean: Tensor, rvar: Tensor, training: bool, eps: float,
              momentum: float) -> Tensor:
    lib = _lib.load()
    B, C, L = y.shape
    stats = torch.empty(4 * C, device=y.device, dtype=torch.float32)
    part, n_slots = None, 0
    if training:
        # (count, shift, Σ(x−shift), Σ(x−shift)²) per (channel, slot), written by the kernel (no zero fill), merged in slot order by
        # fst_bn_finalize
        part = torch.empty(C, BN_SLOTS, 4, device=y.device, dtype=torch.float32)
        check(lib.fst_bn_stats(ptr(y), B, C, L, ptr(part), y.numel(), stream_ptr()), "fst_bn_stats")
        part = _dist.gather_slots(part)                  # global-batch mode (SyncBN): every rank's slots, in rank order
        n_slots = part.size(1)
    check(lib.fst_bn_finalize(ptr(part), n_slots, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), int(training), C, eps,
                              momentum, ptr(stats), stream_ptr()), "fst_bn_finalize")
    return stats


def _bn_backward(dy: Tensor, y: Tensor, out: Optional[Tensor], stats: Tensor, relu: bool, training: bool,
                 need_dx: bool = True):
    lib = _lib.load()
    B, C, L = y.shape
    # per-slot partial sums [2][C][slots] (Σ dy', Σ dy'·x̂), written by the kernel; fst_bn_bwd_apply adds the slots in order and
    # leaves the totals (dβ | dγ) in ``red`` — no zero fill, no atomics, no reduction launch
    part = torch.empty(2, C, BN_SLOTS, device=y.device, dtype=torch.float32)
    check(lib.fst_bn_bwd_reduce(ptr(dy), ptr(y), ptr(out), ptr(stats), B, C, L, int(relu), ptr(part), _same_numel(dy, y, out),
                                stream_ptr()),
          "fst_bn_bwd_reduce")
    if not need_dx:
        red = part.sum(dim=2).view(2 * C)
        return None, red[C:], red[:C]
    dx = torch.empty_like(y)
    # Σ_t dx per (sample, channel), left by the same launch: the conv in front of this BatchNorm takes its bias gradient from them
    # (ConvFn.backward) instead of a pass of its own over dx
    rs = torch.empty(B, C, device=y.device, dtype=torch.float32)
    if training and _dist.global_batch_active():
        # the two batch means of the backward formula run over every rank's samples; the parameter gradients
        # (returned below) stay local sums — the gradient bucket averages them like every other parameter
        red = part.sum(dim=2).view(2 * C)
        red_g = red.clone()
        B_total = _dist.sum_over_ranks_(red_g) * B
        check(lib.fst_bn_bwd_apply(ptr(dy), ptr(y), ptr(out), ptr(stats), ptr(red_g), 1, None, ptr(dx), ptr(rs), B, C, L, int(relu),
                                   int(training), B_total, _same_numel(dy, y, out, dx), stream_ptr()), "fst_bn_bwd_apply")
    else:
        red = torch.empty(2 * C, device=y.device, dtype=torch.float32)
        check(lib.fst_bn_bwd_apply(ptr(dy), ptr(y), ptr(out), ptr(stats), ptr(part), BN_SLOTS, ptr(red), ptr(dx), ptr(rs), B, C, L,
                                   int(relu), int(training), B, _same_numel(dy, y, out, dx), stream_ptr()), "fst_bn_bwd_apply")
    _ROW_SUMS.attach(dx, rs)
    return dx, red[C:], red[:C]                                   # dx, dgamma, dbeta


class BNActFn(torch.autograd.Function):
    """BatchNorm1d (train: batch stats + running update; eval: running stats) → optional ReLU."""

    @staticmethod
    def forward(ctx, y, gamma, beta, rmean, rvar, training: bool, relu: bool, eps: float, momentum: float):
        lib = _lib.load()
        y = y.contiguous()
        B, C, L = y.shape
        stats = _bn_stats(y, gamma, beta, rmean, rvar, training, eps, momentum)
        out = torch.empty_like(y)
        check(lib.fst_bn_apply(ptr(y), ptr(stats), None, None, ptr(out), B, C, L, int(relu), _same_numel(y, out), stream_ptr()),
              "fst_bn_apply")
        ctx.save_for_backward(y, out, stats)
        ctx.relu, ctx.training = relu, training
        return out

    @staticmethod
    def backward(ctx, dout):
        y, out, stats = ctx.saved_tensors
        dx, dg, db = _bn_backward(dout.contiguous(), y, out if ctx.relu else None, stats, ctx.relu, ctx.training,
                                  ctx.needs_input_grad[0])
        return dx, dg, db, None, None, None, None, None, None


class BNAddBNReluFn(torch.autograd.Function):
    """relu(BN_a(ya) + BN_b(yb)) in one pass — the residual join of Res_OS_layer (OS_CNN.py:176-180)."""

    @staticmethod
    def forward(ctx, ya, ga, ba, rma, rva, yb, gb, bb, rmb, rvb, training: bool, eps: float, momentum: float):
        lib = _lib.load()
        ya, yb = ya.contiguous(), yb.contiguous()
        B, C, L = ya.shape
        sa = _bn_stats(ya, ga, ba, rma, rva, training, eps, momentum)
        sb = _bn_stats(yb, gb, bb, rmb, rvb, training, eps, momentum)
        out = torch.empty_like(ya)
        check(lib.fst_bn_apply(ptr(ya), ptr(sa), ptr(yb), ptr(sb), ptr(out), B, C, L, 1, _same_numel(ya, yb, out), stream_ptr()),
              "fst_bn_apply")
        ctx.save_for_backward(ya, yb, out, sa, sb)
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, dout):
        ya, yb, out, sa, sb = ctx.saved_tensors
        dout = dout.contiguous()
        dxa, dga, dba = _bn_backward(dout, ya, out, sa, True, ctx.training, ctx.needs_input_grad[0])
        dxb, dgb, dbb = _bn_backward(dout, yb, out, sb, True, ctx.training, ctx.needs_input_grad[5])
        return dxa, dga, dba, None, None, dxb, dgb, dbb, None, None, None, None, None


# --------------------------------------------------------------------------------------------------
# WaveGlow: WN stack, affine coupling
# --------------------------------------------------------------------------------------------------
class WNSpecs:
    """Conv specs of one WN(n_in=h, n_layers, n_channels=n, k=3) (Simplified_NF_WaveGlow.py:55-99)."""

    def __init__(self, h: int, n: int, n_layers: int = 8, kernel: int = 3):
        self.h, self.n, self.n_layers, self.kernel = h, n, n_layers, kernel
        self.start = ConvSpec(n, h)
        self.ins = [ConvSpec(2 * n, n, kernel, 2 ** i, int((kernel * 2 ** i - 2 ** i) / 2), C1=h) for i in range(n_layers)]
        self.rs = [ConvSpec(2 * n if i < n_layers - 1 else n, n) for i in range(n_layers)]
        self.end = ConvSpec(2 * h, n)
        # dacts = W_rsᵀ·[d_a ; d_out]: rows n, K = 2n channels from two tensors
        self.rs_T = build_plan(n, [Segment(0, n, 0, 1), Segment(1, n, 0, 1)], 1, 1, 0, chunk_c=PIPE_C)
        self.rs_T_last = build_plan(n, [Segment(0, n, 0, 1)], 1, 1, 0, chunk_c=PIPE_C)
        # The effective weights travel as ONE flat tensor (WNFn's third argument): start_w, start_b, cond_w, cond_b, end_w,
        # end_b, in_w[0..], in_b[0..], rs_w[0..], rs_b[0..].  The three applications of a WN in a train step (two forward
        # flows and infer) then hand autograd three flat gradients to sum — two adds per WN instead of two per tensor
        # (228 tiny launches per step) — and the backward writes every gradient straight into its segment.
        nl = n_layers
        self.shapes = ([(n, h, 1), (n,), (2 * n * nl, h, 1), (2 * n * nl,), (2 * h, n, 1), (2 * h,)]
                       + [(2 * n, n, kernel)] * nl + [(2 * n,)] * nl
                       + [(2 * n if i < nl - 1 else n, n, 1) for i in range(nl)] + [(2 * n if i < nl - 1 else n,) for i in range(nl)])
        self.offsets = [0]
        for sh in self.shapes:
            self.offsets.append(self.offsets[-1] + int(math.prod(sh)))
        self.flat_numel = self.offsets[-1]

    def flatten(self, weights: Sequence[Tensor]) -> Tensor:
        assert len(weights) == len(self.shapes) and all(tuple(w.shape) == sh for w, sh in zip(weights, self.shapes))
        return torch.cat([w.reshape(-1) for w in weights])

    def unflatten(self, flat: Tensor) -> List[Tensor]:
        assert flat.numel() == self.flat_numel and flat.is_contiguous()
        return [flat[self.offsets[i]: self.offsets[i + 1]].view(sh) for i, sh in enumerate(self.shapes)]


def wn_wgrad_ok(kind: int, B: int, L: int, n: int, h: int, dil: int, a: Optional[Tensor] = None) -> bool:
    """Whether the time-as-k weight-gradient kernels of csrc/wn_wgrad.hip serve this layer (kind 0 = in_layer + cond_layer,
    1 = res_skip): split-bf16 arithmetic, L % 32 == 0, n < 128, h <= 32; tap shifts that are not multiples of 4 samples
    (dilation 1, 2) need 16 readable bytes either side of ``a`` (``empty_with_slack``).  Everything else stays on fst_conv_wgrad."""
    if MATH != "bf16x3" or os.environ.get("FST_WN_WGRAD", "1") == "0":
        return False
    served = _lib.load().fst_wn_wgrad_ok(kind, B, L, n, h, dil)
    return served == 1 or (served == 2 and a is not None and has_slack(a))


def empty_with_slack(B: int, C: int, L: int, device) -> Tensor:
    """A contiguous [B, C, L] fp32 tensor with 16 readable bytes in front of and behind it inside its own allocation (the
    16-byte LDS-DMA pieces of a tap shifted by 1-3 samples start up to 3 samples outside a row: fst_wn_wgrad_in, a_slack)."""
    buf = torch.empty(B * C * L + 8, device=device, dtype=torch.float32)
    return buf[4: 4 + B * C * L].view(B, C, L)


def has_slack(t: Tensor) -> bool:
    off = t.storage_offset()
    return t.is_contiguous() and off >= 4 and t.untyped_storage().nbytes() >= (off + t.numel() + 4) * 4


WN_WGRAD_MAX_SETS = 3        # (WW_MAX_SETS of csrc/wn_wgrad.hip)


def _sets(x) -> list:
    return list(x) if isinstance(x, (list, tuple)) else [x]


def wn_wgrad_in(dg, a, u0, dw_in: Tensor, dw_cond: Tensor, n: int, h: int, dil: int) -> None:
    """dw_in [2n, n, 3] = Σ dg ⊗ a(t + (τ−1)·dil),  dw_cond [2n, h, 1] = Σ dg ⊗ u0 — written in place (fst_wn_wgrad_in).
    ``dg``, ``a``, ``u0``: one tensor each, or equally long lists of up to 3 same-shaped operand sets whose gradients are summed
    by the one launch (the applications of a WN in a train step)."""
    lib = _lib.load()
    dgs, as_, u0s = _sets(dg), _sets(a), _sets(u0)
    if not (1 <= len(dgs) <= WN_WGRAD_MAX_SETS and len(as_) == len(dgs) and len(u0s) == len(dgs)):
        raise ValueError(f"wn_wgrad_in: {len(dgs)}/{len(as_)}/{len(u0s)} operand sets (1..{WN_WGRAD_MAX_SETS}, equally many)")
    B, _, L = dgs[0].shape
    u0_bs, _ = _ncl(u0s[0], "u0")
    numel = _same_numel(*as_)
    slack = True
    for dg_s, a_s, u0_s in zip(dgs, as_, u0s):
        if dg_s.numel() != 2 * numel or not dg_s.is_contiguous() or tuple(a_s.shape) != (B, n, L) or tuple(u0_s.shape) != (B, h, L):
            raise ValueError("wn_wgrad_in: dg must be contiguous [B, 2n, L], a [B, n, L], u0 [B, h, L]")
        if _ncl(u0_s, "u0")[0] != u0_bs:
            raise ValueError("wn_wgrad_in: the operand sets' u0 tensors must share one batch stride")
        slack = slack and has_slack(a_s)
    for t, sh in ((dw_in, (2 * n, n, 3)), (dw_cond, (2 * n, h, 1))):
        if tuple(t.shape) != sh or not t.is_contiguous() or t.dtype != torch.float32:
            raise ValueError(f"wn_wgrad_in: gradient target of shape {tuple(t.shape)}, expected contiguous {sh}")
    ws_n = lib.fst_wn_wgrad_workspace_floats(0, B, L, n, h, 0)
    ws = torch.empty(ws_n, device=dgs[0].device, dtype=torch.float32)
    ns = len(dgs)
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_wn_wgrad_in(_ptr_sets(dgs), _ptr_sets(as_), _ptr_sets(u0s), ns, u0_bs, ptr(dw_in), ptr(dw_cond), ptr(ws), ws_n, B, L,
                              n, h, dil, int(slack), numel, stream_ptr()), "fst_wn_wgrad_in")
    if t0 is not None:
        KERNEL_TIMER.end("wn_wgrad_kernel<2, 3> bf3", t0, 2.0 * ns * B * L * 2 * n * (3 * n + h), 4.0 * ns * B * L * (2 * n + n + h))


def wn_wgrad_rs(d_a, d_out, ts, dw_rs: Tensor, last: bool, n: int) -> None:
    """dw_rs [2n | n, n, 1] = Σ [d_a ; d_out] ⊗ (t·s), acts = t·s re-formed from the saved gate halves ts [B, 2n, L] (fst_wn_wgrad_rs).
    One tensor each or lists of up to 3 operand sets, as ``wn_wgrad_in``; ``d_a`` is None exactly on the last layer."""
    lib = _lib.load()
    d_outs, tss = _sets(d_out), _sets(ts)
    d_as = [None] * len(d_outs) if d_a is None else _sets(d_a)
    if not (1 <= len(d_outs) <= WN_WGRAD_MAX_SETS and len(tss) == len(d_outs) and len(d_as) == len(d_outs)):
        raise ValueError(f"wn_wgrad_rs: {len(d_as)}/{len(d_outs)}/{len(tss)} operand sets (1..{WN_WGRAD_MAX_SETS}, equally many)")
    B, _, L = d_outs[0].shape
    numel = _same_numel(*d_outs, *[t for t in d_as if t is not None])
    M = n if last else 2 * n
    for da_s, ts_s in zip(d_as, tss):
        if ts_s.numel() != 2 * numel or not ts_s.is_contiguous() or (da_s is None) != bool(last):
            raise ValueError("wn_wgrad_rs: ts must be contiguous [B, 2n, L]; d_a is None exactly on the last layer")
    if tuple(dw_rs.shape) != (M, n, 1) or not dw_rs.is_contiguous() or dw_rs.dtype != torch.float32:
        raise ValueError(f"wn_wgrad_rs: gradient target of shape {tuple(dw_rs.shape)}, expected contiguous {(M, n, 1)}")
    ws_n = lib.fst_wn_wgrad_workspace_floats(1, B, L, n, 0, int(last))
    ws = torch.empty(ws_n, device=d_outs[0].device, dtype=torch.float32)
    ns = len(d_outs)
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_wn_wgrad_rs(None if last else _ptr_sets(d_as), _ptr_sets(d_outs), _ptr_sets(tss), ns, ptr(dw_rs), ptr(ws), ws_n,
                              int(last), B, L, n, numel, stream_ptr()), "fst_wn_wgrad_rs")
    if t0 is not None:
        KERNEL_TIMER.end("wn_wgrad_kernel<2, 2> bf3", t0, 2.0 * ns * B * L * M * n, 4.0 * ns * B * L * (M + 2 * n))


def dense_tap_wgrad(dy: Tensor, x: Tensor, dw: Tensor, M: int, C: int, K: int, pad_left: int) -> None:
    """dw [M, C, K] = Σ_{b,t} dy[b, m, t]·x[b, c, t + k − pad_left] for every tap k < K <= 96 — written in place (fst_dense_tap_wgrad)."""
    lib = _lib.load()
    B, _, L = dy.shape
    if tuple(dy.shape) != (B, M, L) or tuple(x.shape) != (B, C, L) or not (dy.is_contiguous() and x.is_contiguous()):
        raise ValueError(f"dense_tap_wgrad: dy {tuple(dy.shape)} / x {tuple(x.shape)} must be contiguous [B, {M}, L] / [B, {C}, L]")
    if tuple(dw.shape) != (M, C, K) or not dw.is_contiguous() or dw.dtype != torch.float32:
        raise ValueError(f"dense_tap_wgrad: gradient target of shape {tuple(dw.shape)}, expected contiguous {(M, C, K)}")
    ws_n = lib.fst_dense_tap_wgrad_workspace_floats(B, L, M, C, K)
    if ws_n <= 0:
        raise ValueError(f"dense_tap_wgrad: unsupported shape B={B} L={L} M={M} C={C} K={K}")
    ws = torch.empty(ws_n, device=dy.device, dtype=torch.float32)
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_dense_tap_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(ws), ws_n, B, L, M, C, K, pad_left, dy.numel(), x.numel(), stream_ptr()),
          "fst_dense_tap_wgrad")
    if t0 is not None:
        KERNEL_TIMER.end(f"tz_wgrad_kernel M={M} C={C} bf3", t0, 2.0 * B * L * M * C * K, 4.0 * B * L * (M + C))


def tap_wgrad(dy: Tensor, x: Tensor, dw: Tensor, M: int, C: int, ntaps: int, dil: int, pad_left: int) -> None:
    """dw [M, C, ntaps] = Σ_{b,t} dy[b, m, t]·x[b, c, t + τ·dil − pad_left] — written in place (fst_tap_wgrad: time-as-k kernel)."""
    lib = _lib.load()
    B, _, L = dy.shape
    if tuple(dy.shape) != (B, M, L) or tuple(x.shape) != (B, C, L) or not (dy.is_contiguous() and x.is_contiguous()):
        raise ValueError(f"tap_wgrad: dy {tuple(dy.shape)} / x {tuple(x.shape)} must be contiguous [B, {M}, L] / [B, {C}, L]")
    if tuple(dw.shape) != (M, C, ntaps) or not dw.is_contiguous() or dw.dtype != torch.float32:
        raise ValueError(f"tap_wgrad: gradient target of shape {tuple(dw.shape)}, expected contiguous {(M, C, ntaps)}")
    ws_n = lib.fst_tap_wgrad_workspace_floats(B, L, M, C, ntaps)
    if ws_n <= 0:
        raise ValueError(f"tap_wgrad: unsupported shape B={B} L={L} M={M} C={C} ntaps={ntaps}")
    ws = torch.empty(ws_n, device=dy.device, dtype=torch.float32)
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_tap_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(ws), ws_n, B, L, M, C, ntaps, dil, pad_left, int(has_slack(x)), dy.numel(),
                            x.numel(), stream_ptr()), "fst_tap_wgrad")
    if t0 is not None:
        KERNEL_TIMER.end("tap_wgrad (wn_wgrad_kernel<2, 3>) bf3", t0, 2.0 * B * L * M * C * ntaps, 4.0 * B * L * (M + C))


def _ptr_sets(ts: Sequence[Tensor]):
    """Host array of the operand sets' device addresses (a ``const float* const*`` argument)."""
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError("operand sets must be fp32 device tensors")
        arr[i] = t.data_ptr()
    return arr


class WNGradPool:
    """Weight-gradient operands of the applications of ONE WN whose weights go through ``WGradJoinFn``: every application's
    backward leaves (dg, a, u0) / (d_a, d_out, ts) of each layer here instead of launching its own weight-gradient kernels; the
    join node's backward — which autograd runs after ALL applications that take part in the pass — sums them with one launch per
    layer and kind."""

    def __init__(self):
        self.pending: dict = {}

    def add(self, key, operands) -> None:
        self.pending.setdefault(key, []).append(operands)


class WGradJoinFn(torch.autograd.Function):
    """Identity on the flat weight tensor of a WN; backward adds the deferred in_layer / cond_layer / res_skip weight gradients
    of every application recorded in ``pool`` to the (autograd-summed) gradient of the flat tensor.  The applications zero those
    segments of the gradients they return, so the summed segments are written, not accumulated."""

    @staticmethod
    def forward(ctx, pool: WNGradPool, specs: "WNSpecs", flat: Tensor):
        ctx.pool, ctx.specs = pool, specs
        return flat.view_as(flat)

    @staticmethod
    def backward(ctx, g):
        S, pool = ctx.specs, ctx.pool
        pending, pool.pending = pool.pending, {}
        if not pending:
            return None, None, g
        g = g.contiguous()
        nl, n, h = S.n_layers, S.n, S.h
        dw = S.unflatten(g)
        g_cond_w, g_in_w, g_rs_w = dw[2], dw[6: 6 + nl], dw[6 + 2 * nl: 6 + 3 * nl]
        for (kind, i), sets in pending.items():
            for c0 in range(0, len(sets), WN_WGRAD_MAX_SETS):
                chunk = sets[c0: c0 + WN_WGRAD_MAX_SETS]
                cols = [list(col) for col in zip(*chunk)]
                if kind == 0:
                    t_in, t_cond = g_in_w[i], g_cond_w[2 * n * i: 2 * n * (i + 1)]
                    if c0:
                        t_in, t_cond = torch.empty_like(t_in), torch.empty_like(t_cond)
                    wn_wgrad_in(cols[0], cols[1], cols[2], t_in, t_cond, n, h, 2 ** i)
                    if c0:
                        g_in_w[i].add_(t_in)
                        g_cond_w[2 * n * i: 2 * n * (i + 1)].add_(t_cond)
                else:
                    last = i == nl - 1
                    t_rs = torch.empty_like(g_rs_w[i]) if c0 else g_rs_w[i]
                    wn_wgrad_rs(None if last else cols[0], cols[1], cols[2], t_rs, last, n)
                    if c0:
                        g_rs_w[i].add_(t_rs)
        return None, None, g


class WNFoldPlan:
    """Row table of ``WNFoldFn`` for one WN: which parameter tensors (weight-normed (v, g) pairs and plain tensors, in
    ``WNSpecs.shapes`` order) fill which segment of the flat weight tensor.  The device table holds the parameters' ADDRESSES: it is
    rebuilt (one small host-to-device copy) only when a parameter's storage moves — never in a steady train loop, so a captured
    hipGraph replays it as it is (optimisers update in place)."""

    def __init__(self, specs: "WNSpecs", normed: Sequence[bool]):
        assert len(normed) == len(specs.shapes)
        self.specs, self.normed = specs, list(normed)
        self._key, self._table = None, None
        self.n_rows = sum(sh[0] if nm else 1 for sh, nm in zip(specs.shapes, normed))
        # gradient buffer layout: the input tensors' gradients back to back, in input order (v, g per normed entry)
        self.grad_offsets, off = [], 0
        for sh, nm in zip(specs.shapes, normed):
            numel = int(math.prod(sh))
            self.grad_offsets.append(off)
            off += numel
            if nm:
                self.grad_offsets.append(off)
                off += sh[0]
        self.grad_numel = off

    def n_inputs(self) -> int:
        return len(self.grad_offsets)

    def table(self, tensors: Sequence[Tensor]) -> Tensor:
        import numpy as np
        key = tuple(t.data_ptr() for t in tensors) + (str(tensors[0].device),)
        if key == self._key:
            return self._table
        rows, it, gi = [], iter(tensors), 0
        for i, (sh, nm) in enumerate(zip(self.specs.shapes, self.normed)):
            dst, numel = self.specs.offsets[i], int(math.prod(sh))
            if nm:
                v, g = next(it), next(it)
                M, rowlen = sh[0], numel // sh[0]
                assert v.is_contiguous() and g.is_contiguous() and tuple(v.shape) == tuple(sh) and g.numel() == M
                m = np.arange(M, dtype=np.int64)
                blk = np.stack([v.data_ptr() + 4 * rowlen * m, g.data_ptr() + 4 * m, dst + rowlen * m, np.full(M, rowlen, np.int64),
                                self.grad_offsets[gi] + rowlen * m, self.grad_offsets[gi + 1] + m], axis=1)
                gi += 2
            else:
                t = next(it)
                assert t.is_contiguous() and t.numel() == numel
                blk = np.array([[t.data_ptr(), 0, dst, numel, self.grad_offsets[gi], 0]], dtype=np.int64)
                gi += 1
            rows.append(blk)
        tab = np.concatenate(rows, axis=0)
        assert tab.shape == (self.n_rows, 6)
        self._table = torch.from_numpy(tab).to(tensors[0].device)
        self._key = key
        return self._table


class WNFoldFn(torch.autograd.Function):
    """flat = the effective weights of one WN (``WNSpecs.shapes`` order) from its parameters: g·v/‖v‖ for the weight-normed convs
    (Simplified_NF_WaveGlow.py:69-99, old-style weight_norm over dim 0), plain copies for the biases and the end conv — ONE launch
    (fst_wn_fold_fwd) instead of 18 weight-norm launches and a concatenation; the backward (fst_wn_fold_bwd) is one launch too."""

    @staticmethod
    def forward(ctx, plan: WNFoldPlan, *tensors: Tensor):
        lib = _lib.load()
        assert len(tensors) == plan.n_inputs()
        for t in tensors:
            _lib.require_gpu_tensor(t, "WN parameter")
        tab = plan.table(tensors)
        dev = tensors[0].device
        flat = torch.empty(plan.specs.flat_numel, device=dev, dtype=torch.float32)
        norms = torch.empty(plan.n_rows, device=dev, dtype=torch.float32)
        check(lib.fst_wn_fold_fwd(ptr(tab), plan.n_rows, ptr(flat), ptr(norms), stream_ptr()), "fst_wn_fold_fwd")
        ctx.plan, ctx.tab = plan, tab
        ctx.shapes = [tuple(t.shape) for t in tensors]
        ctx.save_for_backward(norms)
        return flat

    @staticmethod
    def backward(ctx, d_flat):
        lib = _lib.load()
        (norms,) = ctx.saved_tensors
        plan: WNFoldPlan = ctx.plan
        d_flat = d_flat.contiguous()
        dpar = torch.empty(plan.grad_numel, device=d_flat.device, dtype=torch.float32)
        check(lib.fst_wn_fold_bwd(ptr(ctx.tab), plan.n_rows, ptr(d_flat), ptr(norms), ptr(dpar), stream_ptr()), "fst_wn_fold_bwd")
        grads = []
        for off, sh in zip(plan.grad_offsets, ctx.shapes):
            grads.append(dpar[off: off + int(math.prod(sh))].view(sh))
        return (None, *grads)


def wn_fused_ok(n: int, h: int, L: int, *tensors: Tensor, kernel: int = 3) -> bool:
    """Whether a WN layer takes the fused one-launch-per-layer kernels (csrc/wn_fused.hip): split-bf16 arithmetic,
    three taps (the kernels hard-code taps at 0 and ±dil; the reference's WN takes any kernel_size,
    Simplified_NF_WaveGlow.py:60 — other sizes run on the generic conv engine), n < 128 (one spare K row carries the
    biases), 16-byte aligned rows."""
    if MATH != "bf16x3" or kernel != 3 or not (0 < n < 128) or L % 4 != 0 or os.environ.get("FST_WN_FUSED", "1") == "0":
        return False
    return all(t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 for t in tensors)


def wn_pack_layer(in_w: Tensor, cond_w: Tensor, in_b: Tensor, cond_b: Tensor, rs_w: Tensor, rs_b: Tensor, n: int, h: int,
                  last: bool) -> Tensor:
    """One layer's weight image for the fused kernels (cached for the step like every packed weight)."""
    lib = _lib.load()
    if in_w.dim() != 3 or in_w.size(0) != 2 * n or in_w.size(1) != n:
        raise ValueError(f"wn_pack_layer: in_w must be [2n, n, taps], got {tuple(in_w.shape)} for n={n}")
    key = None
    if _PACK_CACHE is not None:
        key = ("wn", n, h, last) + tuple((t.data_ptr(), t._version) for t in (in_w, cond_w, in_b, cond_b, rs_w, rs_b))
        hit = _PACK_CACHE.get(key)
        if hit is not None:
            return hit[0]
    nbytes = lib.fst_wn_image_bytes(n, h)
    img = torch.empty(nbytes // 4, device=in_w.device, dtype=torch.float32)
    src = [t.contiguous() for t in (in_w, cond_w, in_b, cond_b, rs_w, rs_b)]
    check(lib.fst_wn_pack(*[ptr(t) for t in src], n, h, in_w.size(2), int(last), ptr(img), nbytes, stream_ptr()), "fst_wn_pack")
    if key is not None:
        _PACK_CACHE[key] = (img, in_w, cond_w, in_b, cond_b, rs_w, rs_b, src)
    return img


def wn_layer_fwd(a: Tensor, u0: Tensor, img: Tensor, ts: Tensor, acts: Optional[Tensor], a_next: Optional[Tensor], out: Tensor,
                 first: bool, last: bool, n: int, h: int, dil: int) -> None:
    lib = _lib.load()
    a_bs, L = _ncl(a, "a")
    u0_bs, _ = _ncl(u0, "u0")
    B = a.size(0)
    numel = _same_numel(a, out, acts, a_next)
    if ts.numel() != 2 * numel or not ts.is_contiguous():
        raise ValueError("wn_layer_fwd: ts must be the contiguous [B, 2n, L] partner of a")
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_wn_layer_fwd(ptr(a), a_bs, ptr(u0), u0_bs, ptr(img), img.numel() * 4, ptr(ts), ptr(acts), ptr(a_next), ptr(out),
                               int(first), int(last), B, L, n, h, dil, numel, stream_ptr()), "fst_wn_layer_fwd")
    if t0 is not None:
        rs_rows = n if last else 2 * n
        flops = 2.0 * B * L * (2 * n * (3 * n + h) + rs_rows * n)
        rows = n + h + (0 if first else n) + (0 if last else n) + n + 2 * n + (n if acts is not None else 0)
        KERNEL_TIMER.end("wn_layer_fwd_kernel", t0, flops, 4.0 * B * L * rows)


def wn_pack_bwd(rs_w: Tensor, n: int, last: bool, acc_order: bool = False) -> Tensor:
    """``acc_order``: the image of fst_wn_stack_bwd (the d_a stages in the k-order of an accumulator tile's rows)."""
    lib = _lib.load()
    key = None
    if _PACK_CACHE is not None:
        key = ("wn_bwd", n, last, acc_order, rs_w.data_ptr(), rs_w._version)
        hit = _PACK_CACHE.get(key)
        if hit is not None:
            return hit[0]
    nbytes = lib.fst_wn_bwd_image_bytes(n, int(last))
    img = torch.empty(nbytes // 4, device=rs_w.device, dtype=torch.float32)
    src = rs_w.contiguous()
    check(lib.fst_wn_pack_bwd(ptr(src), n, int(last), int(acc_order), ptr(img), nbytes, stream_ptr()), "fst_wn_pack_bwd")
    if key is not None:
        _PACK_CACHE[key] = (img, rs_w, src)
    return img


def wn_bwd_partials(n_layers: int, B: int, L: int, device) -> Tensor:
    """[n_layers, 256, B·⌈L/128⌉] per-workgroup row sums of dg, one slab per layer (written by fst_wn_layer_bwd, every entry)."""
    return torch.empty(n_layers, 256, B * ((L + 127) // 128), device=device, dtype=torch.float32)


def wn_layer_bwd(d_a: Optional[Tensor], d_out: Tensor, ts: Tensor, img: Tensor, dg: Tensor, last: bool, n: int,
                 want_row_sums: bool = False, sums_out: Optional[Tensor] = None, part: Optional[Tensor] = None) -> Optional[Tensor]:
    """``want_row_sums``: also returns Σ_{b,t} dg[:, row, :] ([2n]) — the in_layer / cond_layer bias gradient — from per-workgroup
    partials the kernel leaves behind (no extra pass over dg).  ``part`` (a [256, B·⌈L/128⌉] slab of ``wn_bwd_partials``): the
    partials are left there and NOT reduced here — the caller adds the slabs of all layers with one launch."""
    lib = _lib.load()
    B, _, L = d_out.shape
    deferred = part is not None
    if part is None and want_row_sums:
        part = torch.empty(256, B * ((L + 127) // 128), device=dg.device, dtype=torch.float32)
    numel = _same_numel(d_out, d_a)
    for t in (ts, dg):
        if t.numel() != 2 * numel or not t.is_contiguous():
            raise ValueError("wn_layer_bwd: ts / dg must be contiguous [B, 2n, L]")
    if part is not None and (part.shape != (256, B * ((L + 127) // 128)) or not part.is_contiguous()):
        raise ValueError(f"wn_layer_bwd: partial-sum slab of shape {tuple(part.shape)}")
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_wn_layer_bwd(ptr(d_a), ptr(d_out), ptr(ts), ptr(img), img.numel() * 4, ptr(dg), ptr(part),
                               0 if part is None else part.size(1), int(last), B, L, n, numel, stream_ptr()), "fst_wn_layer_bwd")
    if t0 is not None:
        k = n if last else 2 * n
        KERNEL_TIMER.end("wn_layer_bwd_kernel", t0, 2.0 * B * L * n * k, 4.0 * B * L * (k + 4 * n))
    if part is None or deferred:
        return None
    if sums_out is not None:
        return torch.sum(part[: 2 * n], dim=1, out=sums_out)
    return part.sum(dim=1)[: 2 * n]


def wn_pack_dgrad(in_w: Tensor, cond_w: Tensor, n: int, h: int) -> Tensor:
    lib = _lib.load()
    key = None
    if _PACK_CACHE is not None:
        key = ("wn_dgrad", n, h, in_w.data_ptr(), in_w._version, cond_w.data_ptr(), cond_w._version)
        hit = _PACK_CACHE.get(key)
        if hit is not None:
            return hit[0]
    nbytes = lib.fst_wn_dgrad_image_bytes(n)
    img = torch.empty(nbytes // 4, device=in_w.device, dtype=torch.float32)
    src = (in_w.contiguous(), cond_w.contiguous())
    check(lib.fst_wn_pack_dgrad(ptr(src[0]), ptr(src[1]), n, h, in_w.size(2), ptr(img), nbytes, stream_ptr()), "fst_wn_pack_dgrad")
    if key is not None:
        _PACK_CACHE[key] = (img, in_w, cond_w, src)
    return img


WN_DGRAD_TILE = 512       # time samples per workgroup of fst_wn_layer_dgrad (the library checks the row-sum extent against it)


def wn_dgrad_ok(n: int, h: int, dil: int) -> bool:
    """Whether fst_wn_layer_dgrad serves this layer: the library's own test (two window slots of 512 + 2·dil samples must fit
    the LDS: dilations up to 128, i.e. WN stacks of up to 8 layers).  Deeper stacks (the reference's WN takes any n_layers)
    fall back to the generic data-gradient launch per layer."""
    return bool(_lib.load().fst_wn_dgrad_fits(n, h, dil))


def wn_dgrad_partials(n_layers: int, B: int, L: int, device) -> Tensor:
    """[n_layers, 128, B·⌈L/512⌉] per-workgroup row sums of d_a, one slab per layer (written by fst_wn_layer_dgrad)."""
    return torch.empty(n_layers, 128, B * ((L + WN_DGRAD_TILE - 1) // WN_DGRAD_TILE), device=device, dtype=torch.float32)


def wn_layer_dgrad(dg: Tensor, img: Tensor, d_a: Optional[Tensor], d_u0: Tensor, n: int, h: int, dil: int,
                   want_row_sums: bool = False, sums_out: Optional[Tensor] = None, part: Optional[Tensor] = None):
    """returns d_a_new = d_a + W_inᵀ (*) dg;  d_u0 += W_condᵀ·dg  — one launch (csrc/wn_fused.hip).
    ``want_row_sums``: returns (d_a_new, Σ_{b,t} d_a_new[:, row, :]) — the residual half of the next res_skip bias gradient
    (reduced into ``sums_out`` [n] when given).  ``part`` (a [128, B·⌈L/512⌉] slab of ``wn_dgrad_partials``): the partials are
    left there, not reduced (returns (d_a_new, None)): the caller adds the slabs of all layers with one launch."""
    lib = _lib.load()
    B, _, L = dg.shape
    deferred = part is not None
    if part is None and want_row_sums:
        part = torch.empty(128, B * ((L + WN_DGRAD_TILE - 1) // WN_DGRAD_TILE), device=dg.device, dtype=torch.float32)
    d_a_new = torch.empty(B, n, L, device=dg.device, dtype=torch.float32)
    numel = _same_numel(d_a_new, d_a)
    if dg.numel() != 2 * numel or not dg.is_contiguous():
        raise ValueError("wn_layer_dgrad: dg must be contiguous [B, 2n, L]")
    d_u0_bs, _ = _ncl(d_u0, "d_u0")
    if tuple(d_u0.shape) != (B, h, L):
        raise ValueError(f"wn_layer_dgrad: d_u0 must be [B, h, L] = {(B, h, L)}, got {tuple(d_u0.shape)}")
    if part is not None and (part.shape != (128, B * ((L + WN_DGRAD_TILE - 1) // WN_DGRAD_TILE)) or not part.is_contiguous()):
        raise ValueError(f"wn_layer_dgrad: partial-sum slab of shape {tuple(part.shape)}")
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_wn_layer_dgrad(ptr(dg), ptr(img), img.numel() * 4, ptr(d_a), ptr(d_a_new), ptr(d_u0), ptr(part),
                                 0 if part is None else part.size(1), B, L, n, h, dil, numel, d_u0_bs, stream_ptr()),
          "fst_wn_layer_dgrad")
    if t0 is not None:
        KERNEL_TIMER.end("wn_layer_dgrad_kernel", t0, 2.0 * B * L * 2 * n * (3 * n + h),
                         4.0 * B * L * (2 * n + (n if d_a is not None else 0) + n + 2 * h))
    if part is None:
        return d_a_new
    if deferred:
        return d_a_new, None
    if sums_out is not None:
        return d_a_new, torch.sum(part[:n], dim=1, out=sums_out)
    return d_a_new, part.sum(dim=1)[:n]


def wn_stack_bwd_ok(n: int, h: int, L: int, nl: int) -> bool:
    """Whether the whole backward of a WN stack runs as ONE persistent launch (fst_wn_stack_bwd: sequences of up to 512 samples —
    a 512-sample tile is then the whole sequence and one workgroup walks every layer of its batch element)."""
    if os.environ.get("FST_WN_STACK", "1") == "0":                                # diagnostics: one launch pair per layer
        return False
    return bool(_lib.load().fst_wn_stack_bwd_ok(n, h, L, nl))


def wn_stack_fwd_ok(n: int, h: int, L: int, nl: int, B: int) -> bool:
    """Whether the forward of a WN stack runs as ONE persistent launch (fst_wn_stack_fwd): sequences that are whole 256-sample
    tiles, and at least one batch element per CU (fewer would leave CUs without a workgroup where the per-layer launches spread
    the tiles of a sequence over several)."""
    if os.environ.get("FST_WN_STACK_FWD", "1") == "0":                            # diagnostics: one launch per layer
        return False
    lib = _lib.load()
    return bool(lib.fst_wn_stack_fwd_ok(n, h, L, nl)) and B * (L // 256) >= 2 * 256 and B >= 256


def wn_stack_fwd(a_list: Sequence[Tensor], u0: Tensor, imgs: Sequence[Tensor], ts_list: Sequence[Tensor], out: Tensor, n: int,
                 h: int) -> None:
    """All layers of a WN stack's forward in one launch (csrc/wn_fused.hip, fst_wn_stack_fwd): ``a_list[i]`` = input of layer i
    (``a_list[0]`` = the start conv's output; ``a_list[i + 1]`` is written as layer i's a_next), ``imgs[i]`` = ``wn_pack_layer`` image,
    ``ts_list[i]`` [B, 2n, L] written, ``out`` [B, n, L] the skip sum."""
    lib = _lib.load()
    nl = len(imgs)
    B, L = u0.size(0), u0.size(2)
    u0_bs, _ = _ncl(u0, "u0")
    numel = _same_numel(out, *a_list)
    bs = (ctypes.c_int64 * nl)(*[_ncl(a, "a")[0] for a in a_list])
    for ts in ts_list:
        if ts.numel() != 2 * numel or not ts.is_contiguous():
            raise ValueError("wn_stack_fwd: every ts must be the contiguous [B, 2n, L] partner of a")
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_wn_stack_fwd(_ptr_table(a_list), bs, _ptr_table(imgs), imgs[0].numel() * 4, _ptr_table(ts_list),
                               _ptr_table(list(a_list[1:]) + [None]), ptr(u0), u0_bs, ptr(out), nl, B, L, n, h, numel, stream_ptr()),
          "fst_wn_stack_fwd")
    if t0 is not None:
        flops = 2.0 * B * L * (nl * 2 * n * (3 * n + h) + (2 * nl - 1) * n * n)
        rows = nl * (n + h + 2 * n) + (nl - 1) * (n + 2 * n) + n      # a, u0 in, t,s out; a_next out, out in/out (first layer: out only)
        KERNEL_TIMER.end("wn_stack_fwd_kernel", t0, flops, 4.0 * B * L * rows)


def _ptr_table(ts: Sequence[Optional[Tensor]]):
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def wn_stack_bwd(ts_list: Sequence[Tensor], imgs_b: Sequence[Tensor], imgs_d: Sequence[Tensor], dgs: Sequence[Tensor],
                 da_out: Sequence[Optional[Tensor]], d_out: Tensor, d_u0: Tensor, n: int, h: int,
                 part_b: Optional[Tensor] = None, part_d: Optional[Tensor] = None) -> None:
    """Layers nl-1 .. 0 of a WN stack's backward in one launch (csrc/wn_fused.hip, fst_wn_stack_bwd): per layer
    dg = gate'(t, s)·W_rsᵀ·[d_a ; d_out],  d_a += W_inᵀ (*) dg,  d_u0 += W_condᵀ·dg, the residual cotangent d_a staying in the
    accumulators from layer to layer.  ``imgs_b``: ``wn_pack_bwd(..., acc_order=True)``.  ``da_out[i]`` (the cotangent of layer i's
    input) is written where a tensor is given — ``da_out[0]`` always; the rest are the res_skip weight gradients' operands.
    ``dgs`` entries may alias one scratch tensor when nothing reads dg afterwards (GradNorm's partial passes).  ``part_b``
    [nl, 256, B] / ``part_d`` [nl, 128, B]: per-sequence row sums of dg / da_out (the bias gradients), both or neither."""
    lib = _lib.load()
    nl = len(ts_list)
    B, _, L = d_out.shape
    numel = _same_numel(d_out, *[t for t in da_out if t is not None])
    for t in list(ts_list) + list(dgs):
        if t.numel() != 2 * numel or not t.is_contiguous():
            raise ValueError("wn_stack_bwd: ts / dg must be contiguous [B, 2n, L]")
    d_u0_bs, _ = _ncl(d_u0, "d_u0")
    if tuple(d_u0.shape) != (B, h, L):
        raise ValueError(f"wn_stack_bwd: d_u0 must be [B, h, L] = {(B, h, L)}, got {tuple(d_u0.shape)}")
    if (part_b is None) != (part_d is None):
        raise ValueError("wn_stack_bwd: row sums of both kinds or of neither")
    if part_b is not None and (tuple(part_b.shape) != (nl, 256, B) or tuple(part_d.shape) != (nl, 128, B)
                               or not (part_b.is_contiguous() and part_d.is_contiguous())):
        raise ValueError("wn_stack_bwd: row-sum slabs must be contiguous [nl, 256, B] and [nl, 128, B]")
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    rs_b = None if part_b is None else _ptr_table([part_b[i] for i in range(nl)])
    rs_d = None if part_d is None else _ptr_table([part_d[i] for i in range(nl)])
    check(lib.fst_wn_stack_bwd(_ptr_table(ts_list), _ptr_table(imgs_b), _ptr_table(imgs_d), _ptr_table(dgs), _ptr_table(da_out),
                               rs_b, rs_d, ptr(d_out), ptr(d_u0), d_u0_bs, nl, B, L, n, h, numel, stream_ptr()),
          "fst_wn_stack_bwd")
    if t0 is not None:
        # per layer: GEMM 3 (K = 2n, n on the top layer) + the data gradient.  Algorithmic bytes (every operand once): t,s and
        # d_out read, dg written and read back, d_u0 in/out, plus every d_a tensor that is written
        flops = 2.0 * B * L * (n * (2 * n * nl - n) + nl * 2 * n * (3 * n + h))
        rows = nl * (2 * n + n + 2 * 2 * n + 2 * h) + n * sum(t is not None for t in da_out)
        KERNEL_TIMER.end("wn_stack_bwd_kernel", t0, flops, 4.0 * B * L * rows)


def _wn_forward(specs: WNSpecs, u0: Tensor, flat: Tensor):
    """Forward of the WN stack (WNFn's docstring).  Returns (o, fused, the tensors ``_wn_backward`` needs)."""
    lib = _lib.load()
    S, nl = specs, specs.n_layers
    flat = flat.contiguous()
    weights = S.unflatten(flat)
    start_w, start_b, cond_w, cond_b, end_w, end_b = weights[:6]
    in_w, in_b = weights[6: 6 + nl], weights[6 + nl: 6 + 2 * nl]
    rs_w, rs_b = weights[6 + 2 * nl: 6 + 3 * nl], weights[6 + 3 * nl: 6 + 4 * nl]
    B, _, L = u0.shape
    h, n = S.h, S.n
    # the layer inputs are allocated with 16 bytes of slack either side: the time-as-k weight-gradient kernel reads the taps of
    # dilation 1 and 2 through 16-byte pieces that start up to 3 samples outside a row
    a = S.start.forward(u0, None, start_w, None, start_b, y=empty_with_slack(B, n, L, u0.device))
    a_list, ts_list, acts_list = [a], [], []
    fused = wn_fused_ok(n, h, L, a, u0, kernel=S.kernel)
    if fused:
        # one launch per layer: dilated conv + cond rows → gate in registers → res_skip → residual / skip adds
        out = torch.empty(B, n, L, device=u0.device, dtype=torch.float32)
        cb = cond_b.view(nl, 2 * n)
        stack = wn_stack_fwd_ok(n, h, L, nl, B)
        if stack:
            # ONE launch for all layers (persistent workgroups walk their batch elements through the stack)
            imgs = [wn_pack_layer(in_w[i], cond_w[2 * n * i: 2 * n * (i + 1)], in_b[i], cb[i], rs_w[i], rs_b[i], n, h, i == nl - 1)
                    for i in range(nl)]
            a_list += [empty_with_slack(B, n, L, u0.device) for _ in range(nl - 1)]
            ts_list = [torch.empty(B, 2 * n, L, device=u0.device, dtype=torch.float32) for _ in range(nl)]
            wn_stack_fwd(a_list, u0, imgs, ts_list, out, n, h)
            a = a_list[-1]
        for i in range(0 if stack else nl):
            last = i == nl - 1
            img = wn_pack_layer(in_w[i], cond_w[2 * n * i: 2 * n * (i + 1)], in_b[i], cb[i], rs_w[i], rs_b[i], n, h, last)
            ts = torch.empty(B, 2 * n, L, device=u0.device, dtype=torch.float32)
            a_next = None if last else empty_with_slack(B, n, L, u0.device)
            # acts = t·s is not written: the res_skip weight gradient re-forms it from the saved halves while staging
            wn_layer_fwd(a, u0, img, ts, None, a_next, out, i == 0, last, n, h, 2 ** i)
            ts_list.append(ts)
            if not last:
                a = a_next
                a_list.append(a)
    else:
        out = torch.zeros(B, n, L, device=u0.device, dtype=torch.float32)
        bias_g = torch.stack(list(in_b)) + cond_b.view(nl, 2 * n)
    for i in range(0 if fused else nl):
        g = S.ins[i].forward(a, u0, in_w[i], cond_w[2 * n * i: 2 * n * (i + 1)], bias_g[i])
        acts = torch.empty(B, n, L, device=u0.device, dtype=torch.float32)
        check(lib.fst_gate_fwd(ptr(g), ptr(acts), B, n, L, _gate_numel(acts, g), stream_ptr()), "fst_gate_fwd")
        ts_list.append(g)
        acts_list.append(acts)
        if i < nl - 1:
            a_next = torch.empty_like(a)
            S.rs[i].forward(acts, None, rs_w[i], None, rs_b[i], y=a_next, res=a, y2=out, msplit=n, flags=EPI_ACC2)
            a = a_next
            a_list.append(a)
        else:
            S.rs[i].forward(acts, None, rs_w[i], None, rs_b[i], y=None, y2=out, msplit=0, flags=EPI_ACC2)
    o = S.end.forward(out, None, end_w, None, end_b)
    if fused:
        acts_list = ts_list                                           # placeholders (same count) for the saved-tensor layout
    return o, fused, (u0, out, *a_list, *ts_list, *acts_list, flat)


def _wn_backward(S: WNSpecs, fused: bool, sv, do: Tensor, d_u0: Tensor, need_w: bool, pool: Optional["WNGradPool"] = None):
    """Backward of the WN stack: the input gradient is ACCUMULATED into ``d_u0`` ([B, h, L], possibly a channel-slice view of
    a wider tensor with an explicit batch stride); returns the flat weight gradient (None unless ``need_w``).
    ``pool``: the in_layer / cond_layer / res_skip weight gradients served by the time-as-k kernels are not computed here — their
    operands are left in the pool and their segments of the returned gradient are zero (``WGradJoinFn`` fills them)."""
    lib = _lib.load()
    nl, h, n = S.n_layers, S.h, S.n
    u0, out = sv[0], sv[1]
    a_list, ts_list, acts_list = sv[2: 2 + nl], sv[2 + nl: 2 + 2 * nl], sv[2 + 2 * nl: 2 + 3 * nl]
    flat = sv[2 + 3 * nl]
    weights = S.unflatten(flat)
    start_w, cond_w, end_w = weights[0], weights[2], weights[4]
    in_w, rs_w = weights[6: 6 + nl], weights[6 + 2 * nl: 6 + 3 * nl]
    # every gradient is written into its segment of one flat tensor; every segment is written in full (all WN convs have
    # dense plans, the bias sums are stored, not accumulated), so the tensor needs no zero fill
    B, _, L = u0.shape
    defer_in = [bool(need_w and pool is not None and fused and wn_wgrad_ok(0, B, L, n, h, 2 ** i, a_list[i])) for i in range(nl)]
    defer_rs = [bool(need_w and pool is not None and fused and wn_wgrad_ok(1, B, L, n, h, 2 ** i)) for i in range(nl)]
    deferring = any(defer_in) or any(defer_rs)
    d_flat = (torch.zeros_like(flat) if deferring else torch.empty_like(flat)) if need_w else None
    if need_w and os.environ.get("FST_DEBUG_POISON") == "1":          # tests: an unwritten element shows up as NaN
        d_flat.fill_(float("nan"))
        for i in range(nl):
            if defer_in[i]:
                S.unflatten(d_flat)[6 + i].zero_()
                S.unflatten(d_flat)[2][2 * n * i: 2 * n * (i + 1)].zero_()
            if defer_rs[i]:
                S.unflatten(d_flat)[6 + 2 * nl + i].zero_()
    dw = S.unflatten(d_flat) if need_w else [None] * len(S.shapes)
    g_start_w, g_start_b, g_cond_w, g_cond_b, g_end_w, g_end_b = dw[:6]
    g_in_w, g_in_b = dw[6: 6 + nl], dw[6 + nl: 6 + 2 * nl]
    g_rs_w, g_rs_b = dw[6 + 2 * nl: 6 + 3 * nl], dw[6 + 3 * nl: 6 + 4 * nl]
    do = do.contiguous()
    dev = u0.device

    d_out = S.end.grad_x0(do, end_w)
    if need_w:
        S.end.grad_w(out, None, do, out0=g_end_w)
        row_sum(do, out=g_end_b)
    # Σ_{b,t} d_out: the res_skip bias gradient of the last layer, and the skip half of every other layer's
    d_out_sum = row_sum(d_out, out=g_rs_b[nl - 1]) if need_w else None
    d_a: Optional[Tensor] = None
    d_a_sum: Optional[Tensor] = None      # Σ_{b,t} of the current d_a rows when the fused dgrad kernel left it behind
    # res_skip bias gradients of layers 0..nl-2 = [Σ d_a ; Σ d_out] are consecutive segments: one [nl-1, 2n] view whose
    # second half is the same for every layer (one broadcast) and whose first half is reduced straight into its row by
    # the data-gradient launch of the layer above
    d_rs_b_all = None
    if need_w and nl > 1:
        o0 = S.offsets[6 + 3 * nl]
        d_rs_b_all = d_flat[o0: o0 + (nl - 1) * 2 * n].view(nl - 1, 2 * n)
        d_rs_b_all[:, n:] = d_out_sum
    fused_bwd = fused and os.environ.get("FST_WN_BWD", "fused") == "fused"          # diagnostics: =unfused
    fused_dg = [fused and wn_dgrad_ok(n, h, 2 ** i) and os.environ.get("FST_WN_DGRAD", "fused") == "fused" for i in range(nl)]
    # bias-gradient row sums: every fused launch leaves per-workgroup partials in its slab; ONE reduction per kind adds the
    # slabs of all layers straight into the flat gradient's segments (instead of one reduction launch per layer and kind)
    stack = fused_bwd and all(fused_dg) and wn_stack_bwd_ok(n, h, L, nl)
    if stack:
        # ---- every layer in ONE persistent launch.  The full pass keeps each layer's dg and d_a (operands of the weight
        # gradients); a partial pass (GradNorm: no weight gradients) rewrites one dg and two d_a scratch tensors layer after layer
        imgs_b = [wn_pack_bwd(rs_w[i], n, i == nl - 1, acc_order=True) for i in range(nl)]
        imgs_d = [wn_pack_dgrad(in_w[i], cond_w[2 * n * i: 2 * n * (i + 1)], n, h) for i in range(nl)]
        new = lambda c: torch.empty(B, c, L, device=dev, dtype=torch.float32)
        if need_w:
            dgs = [new(2 * n) for _ in range(nl)]
            da_out = [new(n) for _ in range(nl)]
        else:
            dgs = [new(2 * n)] * nl
            da_out = [new(n)] + [None] * (nl - 1)         # d_a stays in the kernel's accumulators; only layer 0's leaves
        da_in = [da_out[i + 1] if i + 1 < nl else None for i in range(nl)]
        part_b = torch.empty(nl, 256, B, device=dev, dtype=torch.float32) if need_w else None
        part_d = torch.empty(nl, 128, B, device=dev, dtype=torch.float32) if need_w else None
        wn_stack_bwd(ts_list, imgs_b, imgs_d, dgs, da_out, d_out, d_u0, n, h, part_b, part_d)
        if need_w:
            for i in range(nl):
                last = i == nl - 1
                if defer_rs[i]:
                    pool.add((1, i), (da_in[i], d_out, ts_list[i]))
                elif wn_wgrad_ok(1, B, L, n, h, 2 ** i):
                    wn_wgrad_rs(da_in[i], d_out, ts_list[i], g_rs_w[i], last, n)
                elif last:
                    S.rs[i].grad_w(ts_list[i][:, :n], None, d_out, x0_mul_off=n * L, out0=g_rs_w[i])
                else:
                    S.rs[i].grad_w(ts_list[i][:, :n], None, da_in[i], d_out, msplit=n, x0_mul_off=n * L, out0=g_rs_w[i])
                if defer_in[i]:
                    pool.add((0, i), (dgs[i], a_list[i], u0))
                elif wn_wgrad_ok(0, B, L, n, h, 2 ** i, a_list[i]):
                    wn_wgrad_in(dgs[i], a_list[i], u0, g_in_w[i], g_cond_w[2 * n * i: 2 * n * (i + 1)], n, h, 2 ** i)
                else:
                    S.ins[i].grad_w(a_list[i], u0, dgs[i], out0=g_in_w[i], out1=g_cond_w[2 * n * i: 2 * n * (i + 1)])
        d_a, d_a_sum = da_out[0], None
    else:
        part_b = wn_bwd_partials(nl, B, L, dev) if (need_w and fused_bwd) else None
        part_d = wn_dgrad_partials(nl, B, L, dev) if (need_w and all(fused_dg)) else None
    for i in (() if stack else reversed(range(nl))):
        last = i == nl - 1
        # ---- through res_skip: rs rows [0,n) carried d_a, rows [n,2n) (or all n rows when last) carried d_out
        dacts = None if fused_bwd else torch.empty(B, n, L, device=dev, dtype=torch.float32)
        if fused_bwd:
            pass
        elif last:
            bf3 = bf3_ok(S.rs_T_last, L)
            a_pk = pack_weights(S.rs_T_last, n, rs_w[i], (0, 1, n, 0), bf3=bf3)
            conv_gemm(S.rs_T_last, a_pk, d_out, None, None, B, L, n, dacts, nb=S.start.nb_for(B, L, pick_mb(n), 0, 0),
                      bf3=bf3)
        else:
            bf3 = bf3_ok(S.rs_T, L)
            a_pk = pack_weights(S.rs_T, n, rs_w[i], (0, 1, n, 0), rs_w[i], (n * n, 1, n, 0), bf3=bf3)
            conv_gemm(S.rs_T, a_pk, d_a, d_out, None, B, L, n, dacts, nb=S.start.nb_for(B, L, pick_mb(n), 0, 0),
                      bf3=bf3)
        if need_w:
            # fused forward: acts = t·s is re-formed from the saved halves (rows [0,n) and [n,2n) of ts) while staging
            x_rs = ts_list[i][:, :n] if fused else acts_list[i]
            mul = n * L if fused else 0
            if fused and os.environ.get("FST_WN_PROD", "1") == "0":                   # diagnostics: materialise acts
                x_rs, mul = (ts_list[i][:, :n] * ts_list[i][:, n:]).contiguous(), 0
            if defer_rs[i] and mul:
                pool.add((1, i), (None if last else d_a, d_out, ts_list[i]))
            elif fused and mul and wn_wgrad_ok(1, B, L, n, h, 2 ** i):
                wn_wgrad_rs(None if last else d_a, d_out, ts_list[i], g_rs_w[i], last, n)
            elif last:
                S.rs[i].grad_w(x_rs, None, d_out, x0_mul_off=mul, out0=g_rs_w[i])      # (its bias gradient is d_out_sum, in place)
            else:
                S.rs[i].grad_w(x_rs, None, d_a, d_out, msplit=n, x0_mul_off=mul, out0=g_rs_w[i])
            if not last and part_d is None and d_a_sum is None:   # Σ d_a not left behind by a fused data-gradient launch
                row_sum(d_a, out=d_rs_b_all[i, :n])
        # ---- through the gate
        dg = torch.empty(B, 2 * n, L, device=dev, dtype=torch.float32)
        dg_sum = None
        if fused_bwd:
            dg_sum = wn_layer_bwd(None if last else d_a, d_out, ts_list[i], wn_pack_bwd(rs_w[i], n, last), dg, last, n,
                                  want_row_sums=need_w, sums_out=g_in_b[i], part=None if part_b is None else part_b[i])
        else:
            check(lib.fst_gate_bwd(ptr(ts_list[i]), ptr(dacts), ptr(dg), B, n, L, _gate_numel(dacts, dg, ts_list[i]),
                                   stream_ptr()), "fst_gate_bwd")
        if need_w:
            # in_layer weights and the layer's rows of the stacked cond_layer weights, unpacked in place
            if defer_in[i]:
                pool.add((0, i), (dg, a_list[i], u0))
            elif fused and wn_wgrad_ok(0, B, L, n, h, 2 ** i, a_list[i]):
                wn_wgrad_in(dg, a_list[i], u0, g_in_w[i], g_cond_w[2 * n * i: 2 * n * (i + 1)], n, h, 2 ** i)
            else:
                S.ins[i].grad_w(a_list[i], u0, dg, out0=g_in_w[i], out1=g_cond_w[2 * n * i: 2 * n * (i + 1)])
            if dg_sum is None and part_b is None:     # fused: partials left by the backward kernel, reduced below
                row_sum(dg, out=g_in_b[i])
        # ---- into the layer input (residual path + dilated conv) and into the conditioning input
        if fused_dg[i]:
            img_d = wn_pack_dgrad(in_w[i], cond_w[2 * n * i: 2 * n * (i + 1)], n, h)
            if need_w:
                d_a, d_a_sum = wn_layer_dgrad(dg, img_d, d_a, d_u0, n, h, 2 ** i, want_row_sums=True,
                                              sums_out=d_rs_b_all[i - 1, :n] if i >= 1 else g_start_b,
                                              part=None if part_d is None else part_d[i])
            else:
                d_a, d_a_sum = wn_layer_dgrad(dg, img_d, d_a, d_u0, n, h, 2 ** i), None
        else:
            d_a, d_a_sum = S.ins[i].grad_x01(dg, in_w[i], cond_w[2 * n * i: 2 * n * (i + 1)], d_a, d_u0), None
    S.start.grad_x0(d_a, start_w, out=d_u0, flags=EPI_ACC1)
    if need_w:
        S.start.grad_w(u0, None, d_a, out0=g_start_w)
        if part_b is not None:
            # in_layer biases of all layers = consecutive segments: one [nl, 2n] reduction
            o0 = S.offsets[6 + nl]
            torch.sum(part_b[:, : 2 * n, :], dim=2, out=d_flat[o0: o0 + nl * 2 * n].view(nl, 2 * n))
        if part_d is not None:
            # Σ d_a entering layer i (left by the data-gradient launch of layer i) = residual half of res_skip bias i−1, and
            # the start conv's bias gradient for i = 0
            if nl > 1:
                torch.sum(part_d[1:, :n, :], dim=2, out=d_rs_b_all[:, :n])
            torch.sum(part_d[0, :n, :], dim=1, out=g_start_b)
        elif d_a_sum is None:
            row_sum(d_a, out=g_start_b)
        # cond_layer bias = the in_layer biases, stacked (consecutive segments: one copy)
        o0 = S.offsets[6 + nl]
        g_cond_b.copy_(d_flat[o0: o0 + nl * 2 * n])
    return d_flat


class WNFn(torch.autograd.Function):
    """The whole gated dilated-conv stack (:101-123) as one autograd node.

    Forward keeps, per layer, the layer input ``a_i``, the gate halves (tanh, sigmoid) and ``acts``;
    the cond_layer 1x1 is folded into each in_layer GEMM as 25 extra K rows, so the [B, 2n·8, L]
    conditioning tensor of the reference (1 GB at B=256, L=512) is never materialised.
    ``u0`` may be a channel-slice view of a wider tensor (explicit batch stride).
    ``flat``: the effective (weight-norm-folded) weights as one tensor in ``WNSpecs.shapes`` order — start_w, start_b,
    cond_w, cond_b, end_w, end_b, in_w[0..], in_b[0..], rs_w[0..], rs_b[0..] (``WNSpecs.flatten``).
    """

    @staticmethod
    def forward(ctx, specs: WNSpecs, u0: Tensor, flat: Tensor):
        o, fused, saved = _wn_forward(specs, u0, flat)
        ctx.specs, ctx.fused = specs, fused
        ctx.save_for_backward(*saved)
        return o

    @staticmethod
    def backward(ctx, do):
        u0 = ctx.saved_tensors[0]
        d_u0 = torch.zeros(u0.size(0), ctx.specs.h, u0.size(2), device=u0.device, dtype=torch.float32)
        d_flat = _wn_backward(ctx.specs, ctx.fused, ctx.saved_tensors, do, d_u0, ctx.needs_input_grad[2] and _want_weight_grad())
        return None, (d_u0 if ctx.needs_input_grad[1] else None), d_flat


def _coupling_forward(u: Tensor, o: Tensor):
    """(x_next, Σ log_s, Σ x_next²) — the two sums as 0-d views of one reduced pair (fst_coupling_fwd leaves per-workgroup partials)."""
    lib = _lib.load()
    B, C, L = u.shape
    xn = torch.empty_like(u)
    part = torch.empty(lib.fst_coupling_sum_slots(B, C // 2, L), 2, device=u.device, dtype=torch.float32)
    check(lib.fst_coupling_fwd(ptr(u), ptr(o), ptr(xn), B, C // 2, L, _same_numel(u, o, xn), ptr(part), stream_ptr()),
          "fst_coupling_fwd")
    sums = part.sum(dim=0)
    return xn, sums[0], sums[1]


def _scalar_f32(g: Optional[Tensor]) -> Optional[Tensor]:
    return None if g is None else g.contiguous().float()


def _coupling_backward(u: Tensor, o: Tensor, dxn: Optional[Tensor], g_ls: Optional[Tensor], g_sq: Optional[Tensor]):
    """(du [B, 2h, L], d_o) of the forward coupling given the cotangents of x_next and of the two sums (each may be None)."""
    lib = _lib.load()
    B, C, L = u.shape
    if dxn is None and g_ls is None and g_sq is None:
        return torch.zeros_like(u), torch.zeros_like(o)
    du, d_o = torch.empty_like(u), torch.empty_like(o)
    dxn = None if dxn is None else dxn.contiguous()
    g_ls, g_sq = _scalar_f32(g_ls), _scalar_f32(g_sq)
    check(lib.fst_coupling_bwd(ptr(u), ptr(o), ptr(dxn), None, ptr(g_ls), ptr(g_sq), ptr(du), ptr(d_o), B, C // 2, L,
                               _same_numel(u, o, dxn, du, d_o), stream_ptr()), "fst_coupling_bwd")
    return du, d_o


class CouplingFn(torch.autograd.Function):
    """x_next = cat(u0, exp(log_s)·u1 + b) with (b, log_s) = split(o) (:173-178).  Also returns Σ log_s and Σ x_next² (two 0-d
    tensors) taken in the same pass: the full-tensor reductions of WaveGlowLoss (:230-241)."""

    @staticmethod
    def forward(ctx, u, o):
        u, o = u.contiguous(), o.contiguous()
        ctx.save_for_backward(u, o)
        ctx.set_materialize_grads(False)      # an unused output's cotangent stays None (no [B, C, L] zero tensor to fill and read)
        return _coupling_forward(u, o)

    @staticmethod
    def backward(ctx, dxn, g_ls, g_sq):
        u, o = ctx.saved_tensors
        return _coupling_backward(u, o, dxn, g_ls, g_sq)


class FlowFn(torch.autograd.Function):
    """One flow step after its invertible 1x1 conv as ONE autograd node (Simplified_NF_WaveGlow.py:165-178 forward, :186-196
    reverse):  o = WN(x[:, :h]);  x_next = coupling(x, o)  (``inverse``: the inverse coupling).

    Returns (x_next, o, Σ log_s, Σ x_next²) (the sums are None in the inverse direction).  As separate nodes the WN's input
    gradient [B, h, L] reaches ``x`` through a slice: autograd zero-fills a [B, 2h, L] tensor, copies the slice in and adds the
    coupling's input gradient — three passes over the feature-sized tensor per flow and backward pass, plus the zero fill of the
    WN's own d_u0.  Here the coupling backward writes the full-width gradient and the WN backward accumulates into its first h
    channels in place (the data-gradient launches take a batch stride)."""

    @staticmethod
    def forward(ctx, specs: WNSpecs, x: Tensor, flat: Tensor, inverse: bool, pool: Optional[WNGradPool] = None):
        lib = _lib.load()
        ctx.pool = pool
        x = x.contiguous()
        h = specs.h
        assert x.size(1) == 2 * h
        o, fused, saved = _wn_forward(specs, x[:, :h], flat)
        B, C, L = x.shape
        if inverse:
            xn = torch.empty_like(x)
            check(lib.fst_coupling_inv_fwd(ptr(x), ptr(o), ptr(xn), B, h, L, _same_numel(x, o, xn), stream_ptr()), "fst_coupling_inv_fwd")
            s_ls = s_sq = None
            keep = xn                                           # the inverse coupling's backward reads its OUTPUT
        else:
            xn, s_ls, s_sq = _coupling_forward(x, o)
            keep = x
        ctx.specs, ctx.fused, ctx.inverse = specs, fused, inverse
        ctx.save_for_backward(keep, o, *saved)
        ctx.set_materialize_grads(False)
        return xn, o, s_ls, s_sq

    @staticmethod
    def backward(ctx, dxn, d_o_ext, g_ls, g_sq):
        lib = _lib.load()
        keep, o = ctx.saved_tensors[0], ctx.saved_tensors[1]
        sv = ctx.saved_tensors[2:]
        S: WNSpecs = ctx.specs
        B, C, L = keep.shape
        if ctx.inverse:
            if dxn is None:
                dx, d_o = torch.zeros_like(keep), torch.zeros_like(o)
            else:
                dx, d_o = torch.empty_like(keep), torch.empty_like(o)
                dxn = dxn.contiguous()
                check(lib.fst_coupling_inv_bwd(ptr(keep), ptr(o), ptr(dxn), ptr(dx), ptr(d_o), B, S.h, L,
                                               _same_numel(keep, o, dxn, dx, d_o), stream_ptr()), "fst_coupling_inv_bwd")
        else:
            dx, d_o = _coupling_backward(keep, o, dxn, g_ls, g_sq)
        if d_o_ext is not None:                                   # someone differentiated the returned WN output itself
            d_o = d_o + d_o_ext
        need_w = ctx.needs_input_grad[2] and _want_weight_grad()
        d_flat = _wn_backward(S, ctx.fused, sv, d_o, dx[:, : S.h], need_w, ctx.pool)     # accumulates into the first h channels of dx
        return None, (dx if ctx.needs_input_grad[1] else None), d_flat, None, None


class CouplingInvFn(torch.autograd.Function):
    """x_next = cat(x0, (x1 − b)/exp(s)) (:193-196)."""

    @staticmethod
    def forward(ctx, x, o):
        lib = _lib.load()
        x, o = x.contiguous(), o.contiguous()
        B, C, L = x.shape
        xn = torch.empty_like(x)
        check(lib.fst_coupling_inv_fwd(ptr(x), ptr(o), ptr(xn), B, C // 2, L, _same_numel(x, o, xn), stream_ptr()),
              "fst_coupling_inv_fwd")
        ctx.save_for_backward(xn, o)
        return xn

    @staticmethod
    def backward(ctx, dxn):
        lib = _lib.load()
        xn, o = ctx.saved_tensors
        B, C, L = xn.shape
        dx, d_o = torch.empty_like(xn), torch.empty_like(o)
        dxn = dxn.contiguous()
        check(lib.fst_coupling_inv_bwd(ptr(xn), ptr(o), ptr(dxn), ptr(dx), ptr(d_o), B, C // 2, L,
                                       _same_numel(xn, o, dxn, dx, d_o), stream_ptr()), "fst_coupling_inv_bwd")
        return dx, d_o


NT_SLICES = 32        # batch slices of the two-stage batch sums of NoiseTransferFn (fixed: the result does not depend on the device)


def _scalar_arg(r):
    """(device pointer or None, host float) of an accumulation ratio given as a Python number or a 0-d device tensor."""
    if isinstance(r, Tensor):
        if r.dtype != torch.float32 or not r.is_cuda or r.numel() != 1:
            raise ValueError("a device ratio must be one fp32 element")
        return ptr(r), 0.0
    return None, float(r)


class NoiseTransferFn(torch.autograd.Function):
    """out = selu(W·((avg_t + r_t·mean_b z_t) − (avg_s + r_s·mean_b z_s)) + bias) + z_s   (/root/reference/widgets.py:150-167), the
    running sums ``avg_t`` / ``avg_s`` ([C, L], detached state) updated in place — three launches (csrc/widgets.hip).
    ``r_t`` / ``r_s``: Python floats or 0-d fp32 device tensors (a captured step refreshes them between replays)."""

    @staticmethod
    def forward(ctx, z_t, z_s, W, bias, avg_t, avg_s, r_t, r_s):
        lib = _lib.load()
        z_t, z_s = z_t.contiguous(), z_s.contiguous()
        B, C, L = z_s.shape
        N = C * L
        if z_t.shape[1:] != z_s.shape[1:] or tuple(avg_t.shape) != (C, L) or tuple(avg_s.shape) != (C, L) or N % 4:
            raise ValueError(f"NoiseTransferFn: z_t {tuple(z_t.shape)}, z_s {tuple(z_s.shape)}, state {tuple(avg_t.shape)} (C·L % 4 == 0)")
        if not (avg_t.is_contiguous() and avg_s.is_contiguous()):
            raise ValueError("NoiseTransferFn: the running sums must be contiguous (they are updated in place)")
        W2 = W.reshape(C, C).contiguous()
        bias = bias.contiguous()
        if z_t.size(0) != B:
            raise ValueError("NoiseTransferFn: the two batches must be equally large (the caller uses the ATen composition otherwise)")
        dev = z_s.device
        S = min(NT_SLICES, B)
        part = torch.empty(2, S, N, device=dev, dtype=torch.float32)
        check(lib.fst_batch_sum(ptr(z_t), ptr(z_s), ptr(part), B, N, S, stream_ptr()), "fst_batch_sum")
        maps = torch.empty(3, C, L, device=dev, dtype=torch.float32)          # dist, pre, learned
        (pt, ft), (ps, fs) = _scalar_arg(r_t), _scalar_arg(r_s)
        check(lib.fst_noise_transfer_fwd(ptr(part), S, B, pt, ps, ft, fs, ptr(avg_t), ptr(avg_s), ptr(W2), ptr(bias), ptr(maps[0]),
                                         ptr(maps[1]), ptr(maps[2]), C, L, stream_ptr()), "fst_noise_transfer_fwd")
        out = torch.empty_like(z_s)
        check(lib.fst_bcast_add(ptr(out), ptr(z_s), ptr(maps[2]), B, N, stream_ptr()), "fst_bcast_add")
        ctx.save_for_backward(maps, W2, *[r for r in (r_t, r_s) if isinstance(r, Tensor)])
        ctx.ratios = (r_t if not isinstance(r_t, Tensor) else None, r_s if not isinstance(r_s, Tensor) else None)
        ctx.w_shape, ctx.B = W.shape, B
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        maps, W2 = ctx.saved_tensors[0], ctx.saved_tensors[1]
        extra = list(ctx.saved_tensors[2:])
        r_t = ctx.ratios[0] if ctx.ratios[0] is not None else extra.pop(0)
        r_s = ctx.ratios[1] if ctx.ratios[1] is not None else extra.pop(0)
        g = g.contiguous()
        B, C, L = g.shape
        N = C * L
        dev = g.device
        S = min(NT_SLICES, B)
        part = torch.empty(S, N, device=dev, dtype=torch.float32)
        check(lib.fst_batch_sum(ptr(g), None, ptr(part), B, N, S, stream_ptr()), "fst_batch_sum")
        small = torch.empty(2, C, L, device=dev, dtype=torch.float32)          # dpre, dd
        check(lib.fst_noise_transfer_bwd(ptr(part), S, ptr(maps[1]), ptr(W2), ptr(small[0]), ptr(small[1]), C, L, stream_ptr()),
              "fst_noise_transfer_bwd")
        dW = db = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dW = torch.empty(C, C, device=dev, dtype=torch.float32)
            db = torch.empty(C, device=dev, dtype=torch.float32)
            check(lib.fst_noise_transfer_dw(ptr(small[0]), ptr(maps[0]), ptr(dW), ptr(db), C, L, stream_ptr()), "fst_noise_transfer_dw")
            dW = dW.view(ctx.w_shape)
        dz_t = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        dz_s = torch.empty_like(g) if ctx.needs_input_grad[1] else None
        if dz_t is not None or dz_s is not None:
            (pt, ft), (ps, fs) = _scalar_arg(r_t), _scalar_arg(r_s)
            check(lib.fst_noise_transfer_bwd_apply(ptr(g), ptr(small[1]), pt, ps, ft, fs, B, ptr(dz_t), ptr(dz_s), N, stream_ptr()),
                  "fst_noise_transfer_bwd_apply")
        return dz_t, dz_s, dW, db, None, None, None, None


class LogDetFn(torch.autograd.Function):
    """log det W of a square fp32 matrix with torch.logdet's conventions (NaN for det < 0, -inf for det = 0), forward and the
    gradient W^{-T} from ONE single-workgroup launch (fst_logdet_inv) — Simplified_NF_WaveGlow.py:40."""

    @staticmethod
    def forward(ctx, W: Tensor):
        lib = _lib.load()
        _lib.require_gpu_tensor(W, "W")
        W = W.contiguous()
        n = W.size(0)
        assert W.dim() == 2 and W.size(1) == n and W.dtype == torch.float32
        out = torch.empty(2, device=W.device, dtype=torch.float32)
        inv_t = torch.empty_like(W)
        check(lib.fst_logdet_inv(ptr(W), n, ptr(out), ptr(inv_t), stream_ptr()), "fst_logdet_inv")
        ctx.save_for_backward(inv_t)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (inv_t,) = ctx.saved_tensors
        return g * inv_t


def logdet(W: Tensor) -> Tensor:
    """``torch.logdet`` for the flow's 1x1 weights: the one-launch kernel up to 96 channels, stock torch beyond."""
    if W.is_cuda and W.dim() == 2 and W.size(0) <= 96 and W.dtype == torch.float32:
        return LogDetFn.apply(W)
    return torch.logdet(W)


# --------------------------------------------------------------------------------------------------
# CPC InfoNCE
# --------------------------------------------------------------------------------------------------
class CPCNceFn(torch.autograd.Function):
    """nce = −(1/(B·T)) Σ_i Σ_b log_softmax(enc_i·pred_iᵀ)[b, col_off+b]; enc_i[b,c] = feat[b,c,t0+i] read in place.
    ``t0`` is a Python int or a 0-d int32 DEVICE tensor (so a captured hipGraph can vary it between replays).
    ``pred`` is [T, Bc, C] with Bc ≥ B columns (the predictions of every rank in global-batch data parallelism)."""

    @staticmethod
    def forward(ctx, feat: Tensor, pred: Tensor, t0, T: int, col_off: int = 0):
        lib = _lib.load()
        _lib.require_gpu_tensor(feat, "feat")
        feat, pred = feat.contiguous(), pred.contiguous()
        B, C, L = feat.shape
        Bc = pred.size(1)
        assert pred.shape == (T, Bc, C) and 0 <= col_off and col_off + B <= Bc
        t0_dev = t0 if isinstance(t0, torch.Tensor) else None
        t0_host = 0 if t0_dev is not None else int(t0)
        assert t0_dev is None or (t0_dev.dtype == torch.int32 and t0_dev.is_cuda)
        assert t0_host + T <= L
        lse = torch.empty(T, B, device=feat.device, dtype=torch.float32)
        # one partial sum per workgroup, added here in slot order (deterministic; float atomics into one scalar were not)
        acc = torch.empty(lib.fst_cpc_nce_slots(T, B, C, Bc), device=feat.device, dtype=torch.float32)
        n_ws = lib.fst_cpc_workspace_floats(T, B, C, Bc)      # the transposed encodings [T, B, C]; > 256 negatives: per-panel softmax statistics
        ws = torch.empty(n_ws, device=feat.device, dtype=torch.float32) if n_ws else None
        check(lib.fst_cpc_nce_fwd(feat.data_ptr() + 4 * t0_host, 1, C * L, L, ptr(t0_dev), ptr(pred), T, B, C, Bc, col_off,
                                  ptr(lse), ptr(acc), ptr(ws), stream_ptr()), "fst_cpc_nce_fwd")
        ctx.save_for_backward(feat, pred, lse)
        ctx.t0_host, ctx.t0_dev, ctx.T, ctx.col_off = t0_host, t0_dev, T, col_off
        return acc.sum() * (-1.0 / (B * T))

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        feat, pred, lse = ctx.saved_tensors
        B, C, L = feat.shape
        T, t0 = ctx.T, ctx.t0_host
        dfeat = torch.zeros_like(feat)
        dpred = torch.empty_like(pred)
        g = g.contiguous().float()
        check(lib.fst_cpc_nce_bwd(feat.data_ptr() + 4 * t0, 1, C * L, L, ptr(ctx.t0_dev), ptr(pred), ptr(lse), T, B, C,
                                  pred.size(1), ctx.col_off, ptr(g), dfeat.data_ptr() + 4 * t0, ptr(dpred), stream_ptr()),
              "fst_cpc_nce_bwd")
        return dfeat, dpred, None, None, None


# --------------------------------------------------------------------------------------------------
# GRU recurrence (CPC context network)
# --------------------------------------------------------------------------------------------------
class GRULastFn(torch.autograd.Function):
    """h_{t_last} of a one-layer GRU with h0 = 0 given the input projections of every step.

    ``xproj`` [B, S, 3H] = x_t·W_ihᵀ + b_ih (torch's gate order r | z | n); ``t_last`` is a Python int or a 0-d int32
    DEVICE tensor (a captured hipGraph varies it between replays).  Forward and backward are ONE persistent launch each
    (csrc/gru.hip) instead of MIOpen's ~10 launches per time step; the weight gradients of the recurrent matrix are two
    small GEMMs over the saved per-step gate gradients."""

    @staticmethod
    def forward(ctx, xproj: Tensor, w_hh: Tensor, b_hh: Tensor, t_last):
        lib = _lib.load()
        _lib.require_gpu_tensor(xproj, "xproj")
        xproj, w_hh, b_hh = xproj.contiguous(), w_hh.contiguous(), b_hh.contiguous()
        B, S, H3 = xproj.shape
        H = H3 // 3
        t_dev = t_last if isinstance(t_last, torch.Tensor) else None
        t_host = 0 if t_dev is not None else int(t_last)
        assert t_dev is None or (t_dev.dtype == torch.int32 and t_dev.is_cuda)
        h_all = torch.zeros(B, S, H, device=xproj.device, dtype=torch.float32)
        gates = torch.empty(B, S, 4 * H, device=xproj.device, dtype=torch.float32)
        check(lib.fst_gru_fwd(ptr(xproj), ptr(w_hh), ptr(b_hh), ptr(h_all), ptr(gates), ptr(t_dev), t_host, B, S, H, h_all.numel(),
                              stream_ptr()), "fst_gru_fwd")
        if t_dev is not None:
            h_t = h_all.gather(1, t_dev.long().view(1, 1, 1).expand(B, 1, H)).reshape(B, H)
        else:
            h_t = h_all[:, t_host, :].contiguous()
        ctx.save_for_backward(w_hh, h_all, gates)
        ctx.t_dev, ctx.t_host = t_dev, t_host
        return h_t

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        w_hh, h_all, gates = ctx.saved_tensors
        B, S, H = h_all.shape
        dxproj = torch.zeros(B, S, 3 * H, device=dh.device, dtype=torch.float32)
        dgh = torch.zeros(B, S, 3 * H, device=dh.device, dtype=torch.float32)
        check(lib.fst_gru_bwd(ptr(w_hh), ptr(h_all), ptr(gates), ptr(dh.contiguous()), ptr(ctx.t_dev), ctx.t_host, ptr(dxproj),
                              ptr(dgh), B, S, H, h_all.numel(), stream_ptr()), "fst_gru_bwd")
        d_w = d_b = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # dW_hh = Σ_{b,s} dgh[b,s] ⊗ h[b,s−1]  (h_{−1} = 0; steps beyond t_last carry zeros in dgh)
            g2 = dgh[:, 1:, :].reshape(-1, 3 * H)
            if S > 1:
                hp = h_all[:, :-1, :].reshape(-1, H)
                # a [3H, H] product with a reduction of B·(S−1) ≈ 2·10⁴: the K split of fst_gemm fills the chip (the library
                # picks 6 workgroups for it: 130 µs)
                d_w = gemm(g2, True, hp, True) if gemm_ok(g2, hp) and os.environ.get("FST_CPC_GEMM", "1") != "0" else g2.t() @ hp
            else:
                d_w = torch.zeros_like(w_hh)
            d_b = dgh.sum(dim=(0, 1))
        return dxproj, d_w, d_b, None


class LSTM2Fn(torch.autograd.Function):
    """h_n of a one-layer LSTM run for TWO steps on the same input with h0 = c0 = 0 (ProbTransfer, widgets.py:46-55), given
    the input projection ``xproj`` [B, 4H] = x·W_ihᵀ + b_ih + b_hh (torch's gate order i | f | g | o).  One launch each way
    (csrc/lstm2.hip) instead of MIOpen's step-by-step RNN; dW_hh is one small GEMM over the saved step-2 gate gradients."""

    @staticmethod
    def forward(ctx, xproj: Tensor, w_hh: Tensor):
        lib = _lib.load()
        _lib.require_gpu_tensor(xproj, "xproj")
        xproj, w_hh = xproj.contiguous(), w_hh.contiguous()
        B, H4 = xproj.shape
        H = H4 // 4
        assert w_hh.shape == (H4, H)
        h2 = torch.empty(B, H, device=xproj.device, dtype=torch.float32)
        save = torch.empty(B, 11 * H, device=xproj.device, dtype=torch.float32)
        check(lib.fst_lstm2_fwd(ptr(xproj), ptr(w_hh.t().contiguous()), ptr(h2), ptr(save), B, H, xproj.numel(), stream_ptr()),
              "fst_lstm2_fwd")
        ctx.save_for_backward(w_hh, save)
        return h2

    @staticmethod
    def backward(ctx, dh2):
        lib = _lib.load()
        w_hh, save = ctx.saved_tensors
        B, H = save.size(0), save.size(1) // 11
        dxproj = torch.empty(B, 4 * H, device=dh2.device, dtype=torch.float32)
        dpre2 = torch.empty(B, 4 * H, device=dh2.device, dtype=torch.float32)
        check(lib.fst_lstm2_bwd(ptr(w_hh), ptr(save), ptr(dh2.contiguous()), ptr(dxproj), ptr(dpre2), B, H, dxproj.numel(),
                                stream_ptr()), "fst_lstm2_bwd")
        d_w = dpre2.t() @ save[:, 10 * H:] if ctx.needs_input_grad[1] else None       # Σ_b dpre2 ⊗ h1
        return dxproj, d_w


# --------------------------------------------------------------------------------------------------
# CDAN random multilinear map: x[B, D] @ R[D, O] with R fixed
# --------------------------------------------------------------------------------------------------
def nt_gemm_ok(M: int, N: int, K: int, *ts: Tensor) -> bool:
    """Whether ``nt_gemm`` serves C[M, N] = A[M, K]·Bm[N, K]ᵀ: at most 256 rows, K a multiple of 32, 16-byte aligned fp32 operands."""
    return (0 < M <= 256 and K % 32 == 0 and MATH == "bf16x3"
            and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts))


def nt_gemm(A: Tensor, Bm: Tensor, epi: Optional[Tuple[Tensor, Tensor, float]] = None, want_raw: bool = False):
    """C = A·Bmᵀ for row-major A [M, K], Bm [N, K] (fst_nt_gemm: split-bf16 products, K split into slabs added in a fixed order).
    ``epi`` = (p [M, ncls], r1 [ncls, N], scale): C = (A·Bmᵀ)·scale·(p·r1) — RandomLayer's epilogue; ``want_raw``: also return A·Bmᵀ."""
    lib = _lib.load()
    M, K = A.shape
    N = Bm.shape[0]
    if Bm.shape[1] != K or not nt_gemm_ok(M, N, K, A, Bm):
        raise ValueError(f"nt_gemm: A {tuple(A.shape)}, Bm {tuple(Bm.shape)} (M <= 256, K % 32 == 0, contiguous fp32 on the GPU, split-bf16 mode)")
    C = torch.empty(M, N, device=A.device, dtype=torch.float32)
    ws_n = lib.fst_nt_gemm_workspace_floats(M, N, K)
    ws = torch.empty(ws_n, device=A.device, dtype=torch.float32)
    raw = torch.empty_like(C) if (want_raw and epi is not None) else None
    p_, r1, scale = (None, None, 1.0) if epi is None else epi
    if epi is not None and (tuple(p_.shape) != (M, r1.shape[0]) or r1.shape[1] != N or not p_.is_contiguous() or not r1.is_contiguous()):
        raise ValueError("nt_gemm: epilogue operands must be contiguous p [M, ncls], r1 [ncls, N]")
    t0 = KERNEL_TIMER.begin() if KERNEL_TIMER is not None else None
    check(lib.fst_nt_gemm(ptr(A), ptr(Bm), ptr(C), ptr(ws), ws_n, M, N, K, ptr(p_), ptr(r1), 0 if epi is None else r1.shape[0], float(scale),
                          ptr(raw), stream_ptr()), "fst_nt_gemm")
    if t0 is not None:
        KERNEL_TIMER.end("nt_gemm (wn_wgrad_kernel<2, 2>) bf3", t0, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N))
    return (C, raw) if want_raw else C


class RandomLayerFn(torch.autograd.Function):
    """RandomLayer.forward for two inputs (C_DAN.py:18-25):  (x·R₀ / √O) ⊙ (p·R₁)  with fixed Gaussian R₀ [D, O], R₁ [ncls, O] — the
    feature-side product on the matrix cores with the class-side product, the scale and the Hadamard product in its epilogue
    (one pass over the [B, O] result).  ``R0t`` = R₀ᵀ kept contiguous (the forward's operand); the backward's is R₀ itself."""

    @staticmethod
    def forward(ctx, x: Tensor, p: Tensor, R0: Tensor, R0t: Tensor, R1: Tensor, scale: float):
        x, p = x.contiguous(), p.contiguous()
        out, y0 = nt_gemm(x, R0t, (p, R1, scale), want_raw=True)
        ctx.save_for_backward(p, R0, R1, y0)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        p, R0, R1, y0 = ctx.saved_tensors
        g = g.contiguous() * ctx.scale
        dx = dp = None
        if ctx.needs_input_grad[0]:
            dx = nt_gemm(g * (p @ R1), R0)                       # (dy ⊙ s / √O)·R₀ᵀ
        if ctx.needs_input_grad[1]:
            dp = (g * y0) @ R1.t()
        return dx, dp, None, None, None, None


class FixedMatmulFn(torch.autograd.Function):
    """y = x @ R for a fixed (non-trainable) R — RandomLayer's big GEMM (C_DAN.py:21).  R is streamed
    once through LDS as the B operand; x is packed as the A operand; K is split across workgroups."""

    @staticmethod
    def forward(ctx, x: Tensor, R: Tensor, Rt: Tensor):
        x = x.contiguous()
        Bx, D = x.shape
        O = R.shape[1]
        ctx.save_for_backward(Rt)
        ctx.shape = (Bx, D, O)
        return _fixed_matmul(x, R, Bx, D, O)

    @staticmethod
    def backward(ctx, dy):
        (Rt,) = ctx.saved_tensors
        Bx, D, O = ctx.shape
        return _fixed_matmul(dy.contiguous(), Rt, Bx, O, D), None, None


_matmul_plans: Dict[Tuple[int, int], Plan] = {}


def _fixed_matmul(x: Tensor, R: Tensor, M: int, K: int, N: int) -> Tensor:
    """[M,K] @ [K,N] via the conv engine: rows = M, 'channels' = K, 'time' = N."""
    key = (M, K)
    if key not in _matmul_plans:
        _matmul_plans[key] = build_plan(M, [Segment(0, K, 0, 1)], 1, 1, 0, chunk_c=PIPE_C)
    plan = _matmul_plans[key]
    bf3 = bf3_ok(plan, N)
    a = pack_weights(plan, M, x, (0, K, 1, 0), bf3=bf3)
    n_tiles = (N + 127) // 128
    ksplit = max(1, min(plan.n_chunks, 512 // max(1, n_tiles * plan.n_mgroups)))
    y = torch.zeros(1, M, N, device=x.device, dtype=torch.float32)
    conv_gemm(plan, a, R.view(1, K, N), None, None, 1, N, M, y, nb=1, ksplit=ksplit,
              flags=EPI_ATOMIC if ksplit > 1 else 0, bf3=bf3)
    return y.view(M, N)
