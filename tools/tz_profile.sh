#!/bin/bash
# Kernel-only durations of the dense many-tap weight gradient under the cost-removal flags, ON the GPU box from the repo root:
#   bash tools/tz_profile.sh "0 1 2 4 8 7 15"   (TZ_EXP masks: libraries built by tools/build_tz_exp.sh; 0 = the product; program directly after `--`)
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
: > $OUT/tz_exp_kernel_times.txt
for e in ${1:-0 7 15}; do
  W=/tmp/prof_tz_$e; rm -rf $W
  lib=$R/feature_level_style_transfer_for_tsc_amd/libfst_hip.so; [ "$e" = 0 ] || lib=$R/build/exp/libfst_tzexp$e.so
  FST_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $W -- python3 $R/tools/tz_time.py > $OUT/tz_exp_$e.log 2>&1 < /dev/null
  f=$(find $W -name "*kernel_stats.csv" | head -1)
  echo "TZ_EXP=$e" >> $OUT/tz_exp_kernel_times.txt
  if [ -n "$f" ]; then grep -E "tz_" "$f" | cut -d, -f1-4 >> $OUT/tz_exp_kernel_times.txt; else echo "  (no stats file)" >> $OUT/tz_exp_kernel_times.txt; fi
done
cat $OUT/tz_exp_kernel_times.txt
