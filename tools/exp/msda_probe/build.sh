#!/bin/bash
# builds tools/exp/msda_probe/libmsda_probe.so (git-ignored; travels with gpurun)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc msda_probe.hip msda_cell.hip -o libmsda_probe.so
