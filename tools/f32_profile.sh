#!/bin/bash
# Kernel-time table of the joint step under FST_MATH=f32 (the unfused conv-engine path), ON the GPU box from the repo root:
#   bash tools/f32_profile.sh   ->  gpurun_out/r04_f32_kernel_stats.csv
R=$(pwd); cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_f32
FST_MATH=f32 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f32 -- python3 $R/bench.py --plain --steps 3 --warmup 1 > $R/gpurun_out/f32_stats.log 2>&1 < /dev/null
f=$(find /tmp/prof_f32 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/r04_f32_kernel_stats.csv && head -25 "$f" | cut -d, -f1-4 | cut -c1-150
