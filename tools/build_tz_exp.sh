#!/bin/bash
# Diagnostic builds of the library for the dense many-tap weight gradient (timing only; TZ_EXP results are wrong):
#   tools/build_tz_exp.sh 1 2 4 ...   ->  build/exp/libfst_tzexp<N>.so   (N = TZ_EXP mask: 1 no MFMAs, 2 no split pass, 4 no LDS-DMA,
#                                          8 no fragment reads, 16 no slab stores, 32 no stages)
#   tools/build_tz_exp.sh stamps      ->  build/exp/libfst_tzstamps.so   (per-phase s_memtime sums: tools/tz_timeline.py)
# use with FST_HIP_LIB=build/exp/<lib> python tools/tz_time.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
for n in "$@"; do
  if [ "$n" = stamps ]; then defs="-DTZ_STAMPS"; name=tzstamps; else defs="-DTZ_EXP=$n"; name=tzexp$n; fi
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-pass-failed $defs -c \
      feature_level_style_transfer_for_tsc_amd/csrc/wn_wgrad.hip -o build/exp/wn_wgrad_$name.o &&
    hipcc --offload-arch=gfx950 -fPIC -shared $(ls build/obj/*.o | grep -v '/wn_wgrad\.o$') \
      build/exp/wn_wgrad_$name.o -o build/exp/libfst_$name.so ) &
done
wait
