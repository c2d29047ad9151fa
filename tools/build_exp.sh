#!/bin/bash
# Diagnostic builds of the library with one cost removed from the bf3 kernel (timing only, results are wrong):
#   tools/build_exp.sh 1 2 4 8 16   ->  build/exp/libfst_exp<N>.so ; use with FST_HIP_LIB=... python tools/wn_micro.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
for n in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-pass-failed -DFST_EXP=$n \
    feature_level_style_transfer_for_tsc_amd/csrc/*.hip -o build/exp/libfst_exp$n.so &
done
wait
