#!/bin/bash
# Diagnostic builds of the library with one cost removed from a fused WN kernel (timing only, results are wrong):
#   tools/build_wn_exp.sh 1 2 4 ...        ->  build/exp/libfst_wnexp<N>.so   (N = WN_EXP mask, forward kernel)
#   DGMODE=1 tools/build_wn_exp.sh 1 8 ... ->  build/exp/libfst_dgexp<N>.so   (N = DG_EXP mask, data-gradient kernel)
# use with FST_HIP_LIB=build/exp/<lib> python tools/wn_fused_time.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
for n in "$@"; do
  if [ -n "$DGMODE" ]; then defs="-DDG_EXP=$n"; name=dgexp$n; else defs="-DWN_EXP=$n"; name=wnexp$n; fi
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-pass-failed $defs -c \
      feature_level_style_transfer_for_tsc_amd/csrc/wn_fused.hip -o build/exp/wn_fused_$name.o &&
    hipcc --offload-arch=gfx950 -fPIC -shared $(ls build/obj/*.o | grep -v '/wn_fused\.o$') \
      build/exp/wn_fused_$name.o -o build/exp/libfst_$name.so ) &      # every object of the library except the one replaced
done
wait
