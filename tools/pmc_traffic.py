#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
into HBM bytes per launch per kernel:  bytes = 2 x FETCH_SIZE x 1024  (gfx950 counts a wide streaming read at half
its bytes)  +  WRITE_SIZE x 1024.   usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> <out.csv> [source note]
With an existing <out.json> the kernels of this run are merged into it (several micro-benchmarks, one summary)."""
import csv, glob, json, re, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            acc[name][0] += 1
            acc[name][1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out, rows = {}, []
for k in sorted(set(fetch) | set(write)):
    if not k.startswith(("conv_", "gate_", "bn_", "pack", "row_sum", "cpc_", "coupling", "wn_", "tz_", "noise_", "batch_sum", "logdet")):
        continue
    nf, sf = fetch.get(k, [0, 0.0]); nw, sw = write.get(k, [0, 0.0])
    rd = 2.0 * 1024.0 * sf / max(nf, 1); wr = 1024.0 * sw / max(nw, 1)
    out[k] = rd + wr
    rows.append((k, nf, rd, wr, rd + wr))
import os
note = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, launch-weighted mean per kernel, FETCH_SIZE x2 correction for gfx950"
merged = {"source": note, "hbm_bytes_per_launch": {}}
if os.path.exists(sys.argv[3]):
    merged = json.load(open(sys.argv[3]))
    merged["source"] = merged["source"] + " | " + note if note not in merged["source"] else merged["source"]
merged["hbm_bytes_per_launch"].update(out)
json.dump(merged, open(sys.argv[3], "w"), indent=1)
with open(sys.argv[4], "w") as f:
    f.write("kernel,launches,read_bytes_per_launch(2x FETCH_SIZE KiB),write_bytes_per_launch,total_bytes_per_launch\n")
    for r in sorted(rows, key=lambda r: -r[4] * r[1]):
        f.write(f"\"{r[0]}\",{r[1]},{r[2]:.0f},{r[3]:.0f},{r[4]:.0f}\n")
print(open(sys.argv[4]).read())
