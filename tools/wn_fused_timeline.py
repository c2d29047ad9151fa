#!/usr/bin/env python3
"""Diagnostics: where a wave of wn_layer_fwd_kernel spends its cycles (needs the FST_STAMPS build:
tools/build_stamps.sh, then FST_HIP_LIB=build/exp/libfst_hip_stamps.so python tools/wn_fused_timeline.py)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import _lib, ops

lib = _lib.load()
lib.fst_debug_wn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
B, L, n, h = 256, 512, 120, 25
names = ["prologue", "vmcnt wait", "barrier", "dma issue", "B frags", "A+mfma", "gate+ts store", "GEMM 2", "epilogue", "whole wave"]
torch.manual_seed(0)
r = lambda *s, k=1.0: torch.randn(*s, device=dev) * k
a, u0 = r(B, n, L), r(B, 2 * h, L)[:, :h]
img = ops.wn_pack_layer(r(2 * n, n, 3, k=.05), r(2 * n, h, 1, k=.1), r(2 * n, k=.1), r(2 * n, k=.1), r(2 * n, n, 1, k=.09), r(2 * n, k=.1), n, h, False)
ts, acts, an, out = torch.empty(B, 2 * n, L, device=dev), torch.empty(B, n, L, device=dev), torch.empty(B, n, L, device=dev), r(B, n, L)
for dil in (1, 8, 128):
    for it in range(3):
        lib.fst_debug_wn_stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.wn_layer_fwd(a, u0, img, ts, None, an, out, False, False, n, h, dil)
        e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    lib.fst_debug_wn_stamps(buf, 1)
    waves = 1024 * 4
    tot = buf[9] / waves
    print(f"== FST_WN_FWD_NW={os.environ.get('FST_WN_FWD_NW', 'auto')} dil {dil}: {e0.elapsed_time(e1) * 1e3:.1f} us, {tot:.0f} cycles/wave (26 + 8 stages)")
    for k in range(9):
        per = buf[k] / waves
        div = 26 if 1 <= k <= 5 else (8 if k == 7 else 1)
        print(f"   {names[k]:14s} {per:9.0f} cyc/wave {100 * per / tot:5.1f}%  {per / div:8.0f} per {'stage' if div > 1 else 'wave'}")
