#!/usr/bin/env python3
"""Print the per-kernel table of a bench.py output file (the roofline leg's HIP-event timings), largest first."""
import json
import sys

for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        ks = d["roofline"]["kernels"]
        print(f"{d['value']:.1f} pairs/s  {d['ms_per_step']:.2f} ms/step   conv-engine kernel time {d['roofline']['conv_engine_ms_per_step']:.2f} ms")
        for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_step"])[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
            print(f"  {k:48s} {v['avg_us']:8.1f} us x {v['launches_per_step']:5.0f} = {v['avg_us'] * v['launches_per_step'] / 1e3:6.2f} ms   mfma {v['frac_of_mfma_peak']:.3f}")
