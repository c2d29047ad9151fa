#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
per kernel the mean counters per launch and the MFMA-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024
SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).   usage: pmc_mfma.py <dir> <out.csv>"""
import csv, glob, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]] += 1
rows = []
for k, c in acc.items():
    if not k.startswith(("conv_", "wn_", "tz_", "cpc_", "gate_", "bn_", "row_sum", "coupling")):
        continue
    n = max(cnt[k].values())
    m = {name: v / max(1, cnt[k][name]) for name, v in c.items()}
    cycles = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cycles * 1024.0) if cycles else 0.0
    rows.append((k, n, cycles, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), busy, m.get("SQ_WAVES", 0.0), m.get("SQ_WAVE_CYCLES", 0.0)))
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches,kernel_cycles(GRBM_GUI_ACTIVE/8),SQ_VALU_MFMA_BUSY_CYCLES,mfma_busy_fraction,SQ_WAVES,SQ_WAVE_CYCLES\n")
    for r in sorted(rows, key=lambda r: -r[2] * r[1]):
        f.write(f"\"{r[0]}\",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.4f},{r[5]:.0f},{r[6]:.0f}\n")
print(open(sys.argv[2]).read())
