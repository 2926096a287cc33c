#!/bin/bash
# Round 6: MSDeformAttn forward with write-through (sc1) output stores vs plain stores: time + HBM traffic + L2 hit rate of the launch
# as the pixel decoder issues it (tools/msda_real.py: 30 frames of 720p).  -> gpurun_out/r06/msda_out_sc1.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for knob in DVIS_MSDA_OUT_SC1=0 DVIS_MSDA_OUT_SC1=1; do
  echo "== $knob"
  for i in 1 2 3; do env $knob python $R/tools/msda_real.py 2>&1 | grep "fused MSDA"; done
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d" " -f1)
    rm -rf /tmp/mt_$n
    env $knob rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/mt_$n -o p -- python $R/tools/msda_real.py > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $(find /tmp/mt_$n -name '*counter_collection.csv' | head -1) | grep -A3 "msda_fwd" | head -4
  done
done
