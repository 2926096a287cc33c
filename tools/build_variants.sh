#!/bin/bash
# Build register-budget variants of the MSDA tile kernel for tools/msda_sweep.py (dev tool).
set -e
cd "$(dirname "$0")/.."
mkdir -p dvis_plus_amd/lib/variants
for w in 2 3 5 6 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden \
    -DDVIS_MSDA_WAVES_PER_SIMD=$w dvis_plus_amd/csrc/common.hip dvis_plus_amd/csrc/msda_forward.hip \
    -o dvis_plus_amd/lib/variants/msda_w$w.so &
done
wait
ls -la dvis_plus_amd/lib/variants
