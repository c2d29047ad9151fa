#!/bin/bash
# Counter / trace evidence of a round, run ON the GPU box from the repo root:  bash tools/profile_round.sh r02
# Separate rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never together with --stats), program
# directly after `--`.  Summaries land in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -e
TAG=${1:-r03}
R=$(pwd)
OUT=$R/gpurun_out
W=/tmp/prof_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp; export TMPDIR=/tmp
export SETS=3            # tools/ww_time.py: one weight-gradient launch sums three operand sets, as the captured train step does
for prog in wn_micro north_star_micro ww_time tz_time; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/${prog}_fetch -- python3 $R/tools/$prog.py > $OUT/${TAG}_${prog}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/${prog}_write -- python3 $R/tools/$prog.py > $OUT/${TAG}_${prog}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d $W/${prog}_mfma -- python3 $R/tools/$prog.py > $OUT/${TAG}_${prog}_mfma.log 2>&1
  TJ=$OUT/traffic_$TAG.json; [ $prog = wn_micro ] || TJ=$OUT/traffic_${TAG}_$prog.json
  python3 $R/tools/pmc_traffic.py $W/${prog}_fetch $W/${prog}_write $TJ $OUT/${TAG}_${prog}_hbm_traffic.csv \
    "$prog.py: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, launch-weighted mean per kernel, FETCH_SIZE x2 (gfx950)" > /dev/null
  python3 $R/tools/pmc_mfma.py $W/${prog}_mfma $OUT/${TAG}_${prog}_mfma_busy.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $W/bench -- python3 $R/bench.py --plain --steps 4 --warmup 1 > $OUT/${TAG}_bench_stats.log 2>&1
cp $(find $W/bench -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
tail -c 400 $OUT/${TAG}_bench_stats.log
