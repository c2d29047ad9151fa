#!/bin/bash
# Diagnostic builds of the library with one cost removed from the fused WN forward kernel (timing only, results are wrong):
#   tools/build_wn_exp.sh 1 2 4 ...  ->  build/exp/libfst_wnexp<N>.so ; use with FST_HIP_LIB=... python tools/wn_fused_time.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
for n in "$@"; do
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-pass-failed -DWN_EXP=$n -c \
      feature_level_style_transfer_for_tsc_amd/csrc/wn_fused.hip -o build/exp/wn_fused_exp$n.o &&
    hipcc --offload-arch=gfx950 -fPIC -shared build/obj/conv_engine.o build/obj/cpc.o build/obj/pointwise.o \
      build/exp/wn_fused_exp$n.o -o build/exp/libfst_wnexp$n.so ) &
done
wait
