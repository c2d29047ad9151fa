"""Host-side conv plans for the HIP conv engine (csrc/conv_engine.hip).

A plan cuts one 1-D convolution  y[b,m,t] = Σ_s Σ_c Σ_tap W_s[m,c,tap] · x_s[b,c,t + tap·dil − pad_left]
into the pieces the f32-MFMA kernels iterate over:

* **M-groups** of ``MB`` 32-row blocks of output rows,
* **chunks** of input channels (a chunk is what one LDS window holds: ≤ ``chunk_cap`` channels of one
  input, optionally restricted to a single tap so that dilated taps do not drag a halo),
* per (M-group, chunk) a **live tap range** and the offset of its **records** in the packed weight
  buffer (one record = the ``MB`` A-operand registers of one ``v_mfma_f32_32x32x2_f32`` k-step),
* for weight-gradient plans an **item table**: 32-row blocks of packed K-rows, four per workgroup.

Omni-scale layers (OS_CNN/OS_CNN.py:46-77) become plans whose tap ranges follow each block's largest
prime kernel, so masked taps are never multiplied (57 % of a dense Kmax conv's MACs at L=512).

Table layout (int32), mirrored by ``plan_view`` in csrc/fst_common.h::

    [0] n_chunks [1] n_mgroups [2] MB [3] ntaps [4] dil [5] pad_left [6] chunk_cap [7] total_records
    [8] n_items  [9] items_per_wg  [10] n_stages  [11..15] reserved
    chunk table   n_chunks × (src, c_begin, c_count, 0)
    (g,q) table   n_mgroups × n_chunks × (tap_lo, tap_hi, record_offset, stage_offset)

``stage_offset`` / ``n_stages`` address the split-bf16 weight image of the pipelined kernel: one stage = one live
(M-group, chunk, tap) triple = one 16-deep ``v_mfma_f32_32x32x16_bf16`` k-step, stored per 32-row block as
64 lanes × 8 bf16 "hi" parts followed by 64 lanes × 8 bf16 "lo" parts (w ≈ hi + lo, each rounded to nearest).
    item table    n_items × (g, q, row_block, 0)          q = −1 marks padding
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

HDR = 16
WG_ITEMS = 4


@dataclass
class Segment:
    """One input tensor's contribution to the K dimension."""
    src: int                                  # 0 or 1: which x pointer
    channels: int
    tap_lo: int
    tap_hi: int
    col_live: Optional[Sequence[Tuple[int, int]]] = None    # per channel live tap range (data-gradient of omni)


@dataclass
class Plan:
    table: np.ndarray
    M: int
    MB: int
    n_mgroups: int
    n_chunks: int
    ntaps: int
    dil: int
    pad_left: int
    total_records: int
    chunk_cap: int
    n_stages: int = 0          # live (M-group, chunk, tap) triples = 16-deep k-steps of the split-bf16 kernel
    _dev: Dict[str, object] = field(default_factory=dict, repr=False)

    @property
    def packed_floats(self) -> int:
        return self.total_records * self.MB * 64

    @property
    def packed_floats_bf3(self) -> int:
        """Size (in 4-byte words) of the split-bf16 image: per stage and 32-row block 64 lanes x (8 hi + 8 lo) bf16."""
        return self.n_stages * self.MB * 512 + 4           # + a 16-byte zero block (source of masked LDS-DMA pieces)

    @property
    def length(self) -> int:
        return int(self.table.size)

    def host_ptr(self) -> int:
        return int(self.table.ctypes.data)

    @property
    def pipeable(self) -> bool:
        """True if fst_conv_gemm dispatches this plan to the software-pipelined kernel (mirrors
        plan_is_pipeable in csrc/conv_engine.hip): single-tap stages of at most 16 channels."""
        en = self.entries()
        return self.chunk_cap <= 16 and bool(((en[:, :, 1] - en[:, :, 0]) <= 1).all())

    @property
    def windowed16(self) -> bool:
        """True for a windowed (multi-tap) plan cut into chunks of 8..16 channels: the split-bf16 window kernel's
        shape (conv_win_bf3_kernel in csrc/conv_engine.hip)."""
        en = self.entries()
        return 8 <= self.chunk_cap <= 16 and bool(((en[:, :, 1] - en[:, :, 0]) > 1).any())

    def dev(self, device):
        """int32 copy of the table on ``device`` (cached)."""
        import torch
        key = str(device)
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(self.table).to(device)
        return self._dev[key]

    # -- views used by tests / the numpy emulation
    def chunks(self) -> np.ndarray:
        return self.table[HDR: HDR + 4 * self.n_chunks].reshape(self.n_chunks, 4)

    def entries(self) -> np.ndarray:
        o = HDR + 4 * self.n_chunks
        return self.table[o: o + 4 * self.n_chunks * self.n_mgroups].reshape(self.n_mgroups, self.n_chunks, 4)

    def items(self) -> np.ndarray:
        o = HDR + 4 * self.n_chunks + 4 * self.n_chunks * self.n_mgroups
        return self.table[o:].reshape(-1, 4)


def pick_mb(M: int) -> int:
    return 1 if M <= 32 else 2 if M <= 64 else 4 if M <= 128 else 8


def build_plan(M: int, segments: Sequence[Segment], ntaps: int, dil: int = 1, pad_left: int = 0,
               MB: Optional[int] = None, chunk_c: int = 32, split_taps: bool = False,
               row_live: Optional[Sequence[Tuple[int, int]]] = None, with_items: bool = False) -> Plan:
    """Build a plan.  ``row_live[m]`` = live tap range of output row m (omni-scale forward);
    ``Segment.col_live[c]`` = live tap range of input channel c (omni-scale data gradient)."""
    if MB is None:
        MB = pick_mb(M)
    assert MB in (1, 2, 4, 8) and M > 0 and ntaps > 0 and dil > 0
    assert chunk_c > 0 and chunk_c % 2 == 0
    rows_per_group = MB * 32
    n_mgroups = (M + rows_per_group - 1) // rows_per_group

    chunks: List[Tuple[int, int, int, int, int, Optional[Sequence[Tuple[int, int]]]]] = []   # src,c0,cnt,lo,hi,col_live
    for seg in segments:
        assert 0 <= seg.tap_lo < seg.tap_hi <= ntaps and seg.channels > 0 and seg.src in (0, 1)
        tap_sets = [(t, t + 1) for t in range(seg.tap_lo, seg.tap_hi)] if split_taps else [(seg.tap_lo, seg.tap_hi)]
        for c0 in range(0, seg.channels, chunk_c):
            cnt = min(chunk_c, seg.channels - c0)
            for lo, hi in tap_sets:
                chunks.append((seg.src, c0, cnt, lo, hi, seg.col_live))
    n_chunks = len(chunks)
    chunk_cap = max((c[2] + 1) & ~1 for c in chunks)

    group_live = []
    for g in range(n_mgroups):
        if row_live is None:
            group_live.append((0, ntaps))
        else:
            rows = row_live[g * rows_per_group: min(M, (g + 1) * rows_per_group)]
            group_live.append((min(r[0] for r in rows), max(r[1] for r in rows)))

    entries = np.zeros((n_mgroups, n_chunks, 4), dtype=np.int32)
    rec = 0
    stage = 0
    items: List[Tuple[int, int, int, int]] = []
    for g in range(n_mgroups):
        n_before = len(items)
        for q, (src, c0, cnt, lo, hi, col_live) in enumerate(chunks):
            lo_q, hi_q = lo, hi
            if col_live is not None:
                cl = col_live[c0: c0 + cnt]
                lo_q, hi_q = max(lo_q, min(r[0] for r in cl)), min(hi_q, max(r[1] for r in cl))
            lo_q, hi_q = max(lo_q, group_live[g][0]), min(hi_q, group_live[g][1])
            if hi_q <= lo_q:
                entries[g, q] = (0, 0, rec, stage)
                continue
            c_pad = (cnt + 1) & ~1
            entries[g, q] = (lo_q, hi_q, rec, stage)
            rec += (hi_q - lo_q) * (c_pad // 2)
            stage += hi_q - lo_q
            if with_items:
                n_rb = ((hi_q - lo_q) * c_pad + 31) // 32
                if hi_q > lo_q + 1:                               # windowed chunk: its own workgroups
                    while (len(items) - n_before) % WG_ITEMS:
                        items.append((g, -1, 0, 0))
                items.extend((g, q, rb, 0) for rb in range(n_rb))
                if hi_q > lo_q + 1:
                    while (len(items) - n_before) % WG_ITEMS:
                        items.append((g, -1, 0, 0))
        if with_items:
            while (len(items) - n_before) % WG_ITEMS:
                items.append((g, -1, 0, 0))
    total_records = rec

    table = np.zeros(HDR + 4 * n_chunks + 4 * n_chunks * n_mgroups + 4 * len(items), dtype=np.int32)
    table[:11] = (n_chunks, n_mgroups, MB, ntaps, dil, pad_left, chunk_cap, total_records, len(items), WG_ITEMS, stage)
    table[HDR: HDR + 4 * n_chunks] = np.array([(c[0], c[1], c[2], 0) for c in chunks], dtype=np.int32).ravel()
    o = HDR + 4 * n_chunks
    table[o: o + entries.size] = entries.ravel()
    if items:
        table[o + entries.size:] = np.array(items, dtype=np.int32).ravel()
    return Plan(table, M, MB, n_mgroups, n_chunks, ntaps, dil, pad_left, total_records, chunk_cap, stage)


# --------------------------------------------------------------------------------------------------
# numpy emulation of the device-side index math (used by CPU tests to validate plans; never by ops)
# --------------------------------------------------------------------------------------------------
def emulate_pack(plan: Plan, w: Sequence[Optional[np.ndarray]], wsrc: Sequence[Tuple[int, int, int, int]]) -> np.ndarray:
    """What ``pack_kernel`` writes.  ``w[s]`` flat weight buffer; ``wsrc[s]`` = (off0, sm, sc, st)."""
    a = np.zeros(plan.packed_floats, dtype=np.float32)
    ch, en, MB = plan.chunks(), plan.entries(), plan.MB
    for g in range(plan.n_mgroups):
        for q in range(plan.n_chunks):
            lo, hi, rec0, _ = en[g, q]
            if hi <= lo:
                continue
            src, c0, cnt, _ = ch[q]
            half_c = ((cnt + 1) & ~1) // 2
            off0, sm, sc, st = wsrc[src]
            for tapi in range(hi - lo):
                for cp in range(half_c):
                    rec = rec0 + tapi * half_c + cp
                    for mb in range(MB):
                        for lane in range(64):
                            m = (g * MB + mb) * 32 + (lane & 31)
                            cl = 2 * cp + (lane >> 5)
                            if m < plan.M and cl < cnt:
                                a[(rec * MB + mb) * 64 + lane] = w[src].flat[off0 + m * sm + (c0 + cl) * sc + (lo + tapi) * st]
    return a


def emulate_conv(plan: Plan, a: np.ndarray, x: Sequence[Optional[np.ndarray]], L: int) -> np.ndarray:
    """What ``conv_gemm_kernel`` accumulates (no bias/epilogue), float64, for one batch item:
    x[s] is [C_s, L]; returns [M, L]."""
    y = np.zeros((plan.n_mgroups * plan.MB * 32, L), dtype=np.float64)
    ch, en, MB = plan.chunks(), plan.entries(), plan.MB
    for g in range(plan.n_mgroups):
        for q in range(plan.n_chunks):
            lo, hi, rec0, _ = en[g, q]
            if hi <= lo:
                continue
            src, c0, cnt, _ = ch[q]
            half_c = ((cnt + 1) & ~1) // 2
            for tapi in range(hi - lo):
                shift = (lo + tapi) * plan.dil - plan.pad_left
                for cp in range(half_c):
                    rec = rec0 + tapi * half_c + cp
                    blk = a[rec * MB * 64: (rec + 1) * MB * 64].reshape(MB, 2, 32)      # [mb][parity][row]
                    for par in range(2):
                        cl = 2 * cp + par
                        if cl >= cnt:
                            continue
                        row = np.zeros(L)
                        t = np.arange(L) + shift
                        ok = (t >= 0) & (t < L)
                        row[ok] = x[src][c0 + cl, t[ok]]
                        for mb in range(MB):
                            r0 = (g * MB + mb) * 32
                            y[r0: r0 + 32] += blk[mb, par][:, None].astype(np.float64) * row[None, :]
    return y[: plan.M]
