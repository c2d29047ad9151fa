#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never shipped, never timed as a product number).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/exp
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-pass-failed -DFST_STAMPS \
  feature_level_style_transfer_for_tsc_amd/csrc/*.hip -o build/exp/libfst_hip_stamps.so
