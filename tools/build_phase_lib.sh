#!/bin/bash
# Builds the library with the rank scatter's phase marks compiled in (tools/phase_timing.py).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/tools/bin"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DNBL_PHASE_TIMING -Wno-unused-function -Wno-pass-failed -shared \
  "$ROOT/nimble_amd/csrc/nbl_kernels.hip" "$ROOT/nimble_amd/csrc/nbl_plan.cpp" "$ROOT/nimble_amd/csrc/nbl_inflate.cpp" \
  -o "$ROOT/tools/bin/libnimble_amd_phase.so" -ldl -Wl,-rpath,/opt/rocm/lib
