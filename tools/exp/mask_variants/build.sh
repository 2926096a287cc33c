#!/bin/bash
# mask_gemm.hip compiled with experiment macros into separate libraries (git-ignored; they travel with gpurun)
cd "$(dirname "$0")"
R=../../../dvis_plus_amd/csrc
for v in base:"" nosched:"-DDVIS_MASK_NOSCHED" b64:"-DDVIS_MASK_B64" b64nosched:"-DDVIS_MASK_B64 -DDVIS_MASK_NOSCHED"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc $flags $R/mask_gemm.hip $R/common.hip -o libmask_$name.so &
done
wait; ls -la libmask_*.so
