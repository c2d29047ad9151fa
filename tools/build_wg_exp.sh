#!/bin/bash
# Diagnostic builds with one cost removed from the split-bf16 weight-gradient kernel (timing only, results are wrong):
#   tools/build_wg_exp.sh 1 2 4 ...  ->  build/exp/libfst_wgexp<N>.so ; use with FST_HIP_LIB=... python tools/wn_micro.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
for n in "$@"; do
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-pass-failed -DWG_EXP=$n -c \
      feature_level_style_transfer_for_tsc_amd/csrc/conv_engine.hip -o build/exp/conv_engine_wg$n.o &&
    hipcc --offload-arch=gfx950 -fPIC -shared build/exp/conv_engine_wg$n.o $(ls build/obj/*.o | grep -v '/conv_engine\.o$') \
      -o build/exp/libfst_wgexp$n.so ) &                                # every object of the library except the one replaced
done
wait
