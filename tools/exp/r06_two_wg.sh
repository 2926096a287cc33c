#!/bin/bash
# Round 6: 256-channel passes of conv1x1_x3 as TWO 4-wave workgroups per CU (DVIS_X3_CONV_TWO_WG = largest C x taps that takes the form)
# vs one 8-wave workgroup: per-shape times of the R50 / pixel-decoder layers.  -> gpurun_out/r06/two_wg.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for v in 0 100000; do
  echo "== DVIS_X3_CONV_TWO_WG=$v"
  DVIS_X3_CONV_TWO_WG=$v timeout 200 python tools/x3_time.py conv 2>&1 | grep "conv " | sed 's/   exact-fp32.*//'
  DVIS_X3_CONV_TWO_WG=$v timeout 200 python tools/x3_time.py conv3 2>&1 | grep "conv3x3" | sed 's/   exact-fp32.*//'
done
