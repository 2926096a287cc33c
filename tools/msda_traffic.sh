#!/bin/bash
# HBM traffic + time of the fused MSDA kernel only (2 PMC passes + timing), A/B over an env knob:  tools/msda_traffic.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
for knob in "${KNOBS[@]:-DVIS_MSDA_VARIANT=0}"; do   # e.g. KNOBS=("DVIS_MSDA_VARIANT=0" "DVIS_MSDA_BOX=1") tools/msda_traffic.sh
  echo "== $knob"
  env $knob python $R/tools/msda_real.py 2>&1 | grep -v ids
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d" " -f1)
    rm -rf /tmp/mt_$n
    env $knob rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/mt_$n -o p -- python $R/tools/msda_real.py > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $(find /tmp/mt_$n -name '*counter_collection.csv' | head -1) | grep -A3 "msda_fwd" | head -4
  done
done
