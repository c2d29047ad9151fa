#!/bin/bash
# Counter evidence for fst_wn_stack_bwd, run ON the GPU box from the repo root: bash tools/profile_stack.sh r04
# (separate rocprofv3 --pmc passes; program directly after `--`).
set -e
TAG=${1:-r04}
R=$(pwd)
OUT=$R/gpurun_out
W=/tmp/prof_stack_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp; export TMPDIR=/tmp
export WS_REPS=3
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/fetch -- python3 $R/tools/wn_stack_time.py > $OUT/${TAG}_stack_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/write -- python3 $R/tools/wn_stack_time.py > $OUT/${TAG}_stack_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $W/l2 -- python3 $R/tools/wn_stack_time.py > $OUT/${TAG}_stack_l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $W/mfma -- python3 $R/tools/wn_stack_time.py > $OUT/${TAG}_stack_mfma.log 2>&1
python3 $R/tools/pmc_traffic.py $W/fetch $W/write $OUT/traffic_${TAG}_stack.json $OUT/${TAG}_stack_hbm_traffic.csv "wn_stack_time.py" > /dev/null
python3 $R/tools/pmc_mfma.py $W/mfma $OUT/${TAG}_stack_mfma_busy.csv
python3 - <<PY
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$W/l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        a = acc[name][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
with open("$OUT/${TAG}_stack_l2_hits.csv", "w") as out:
    out.write("kernel,launches,TCC_HIT_sum per launch,TCC_MISS_sum per launch,hit rate\n")
    for k, v in sorted(acc.items()):
        if not k.startswith("wn_"): continue
        h, m = v["TCC_HIT_sum"], v["TCC_MISS_sum"]
        hh, mm = h[1] / max(h[0], 1), m[1] / max(m[0], 1)
        out.write(f"\"{k}\",{h[0]},{hh:.0f},{mm:.0f},{hh / max(hh + mm, 1):.3f}\n")
print(open("$OUT/${TAG}_stack_l2_hits.csv").read())
PY
cat $OUT/${TAG}_stack_hbm_traffic.csv $OUT/${TAG}_stack_mfma_busy.csv
