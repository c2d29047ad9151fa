#!/usr/bin/env python3
"""Diagnostics: where a wave of tz_wgrad_kernel spends a stage (per-phase s_memtime sums of the stamp build).
   tools/build_tz_exp.sh stamps && FST_HIP_LIB=build/exp/libfst_tzstamps.so python tools/tz_timeline.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feature_level_style_transfer_for_tsc_amd import _lib, ops

lib = _lib.load()
fn = ctypes.CDLL(os.environ["FST_HIP_LIB"]).fst_debug_tz_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
NAMES = ["wait for k-step-0 fragments", "barrier", "k-step-1 + split reads (issue)", "wait for those reads", "MFMA groups ks0", "dy split",
         "LDS-DMA issue", "x split", "waits before the barrier", "next frags + MFMA groups ks1", "whole loop"]
B, L = 256, 512
for (M, C, K) in ((225, 25, 89), (25, 50, 89)):
    x, dy, dw = torch.randn(B, C, L, device="cuda"), torch.randn(B, M, L, device="cuda"), torch.empty(M, C, K, device="cuda")
    run = lambda: ops.dense_tap_wgrad(dy, x, dw, M, C, K, (K - 1) // 2)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    fn(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    fn(buf, 1)
    tot = buf[10]
    print(f"== {M}x{C}x{K}: {1000 * e0.elapsed_time(e1):.1f} us (stamp build, incl. reduce)")
    for i, n in enumerate(NAMES):
        print(f"   {n:28s} {buf[i]:14d}  {100.0 * buf[i] / max(tot, 1):5.1f} %")
