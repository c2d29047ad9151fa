#!/usr/bin/env python3
"""Diagnostics: where a wave of wn_layer_dgrad_kernel spends its cycles (needs the FST_STAMPS build:
tools/build_stamps.sh, then FST_HIP_LIB=build/exp/libfst_hip_stamps.so python tools/wn_dgrad_timeline.py)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import _lib, ops

lib = _lib.load()
lib.fst_debug_wn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
B, L, n, h = 256, 512, 120, 25
names = ["first prologue", "vmcnt wait", "barrier", "dma issue", "B frags+mfma", "hand-over+epilogue", "  of it: slowest wave", "  of it: next prologue"]
torch.manual_seed(0)
r = lambda *s, k=1.0: torch.randn(*s, device=dev) * k
dg, d_a, d_u0 = r(B, 2 * n, L), r(B, n, L), r(B, h, L)
img = ops.wn_pack_dgrad(r(2 * n, n, 3, k=.05), r(2 * n, h, 1, k=.1), n, h)
for dil in (1, 8, 64, 128):
    for it in range(3):
        lib.fst_debug_wn_stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.wn_layer_dgrad(dg, img, d_a, d_u0, n, h, dil)
        e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    lib.fst_debug_wn_stamps(buf, 1)
    tiles = buf[10] / 8                                   # one count per wave per tile
    waves = 8 * min(256, tiles)
    tot = buf[9] / waves
    print(f"== dil {dil}: {e0.elapsed_time(e1) * 1e3:.1f} us, {tot:.0f} cycles/wave, {tiles / (waves / 8):.1f} tiles/workgroup, 15 stages/tile")
    for k in range(8):
        per = buf[k] / waves
        print(f"   {names[k]:20s} {per:9.0f} cyc/wave {100 * per / tot:5.1f}%   {per / (15 * tiles / (waves / 8)) if 1 <= k <= 4 else per:8.0f} per {'stage' if 1 <= k <= 4 else 'wave'}")
