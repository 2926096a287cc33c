#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$R; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r04_msda_lds_dma_probe.txt
timeout 300 python tools/exp/msda_probe/probe.py 0 64 1 2 >> $O/r04_msda_lds_dma_probe.txt 2>&1 < /dev/null
for e in 0 64; do
  for c in "FETCH_SIZE" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    n=$(echo $c | cut -d" " -f1); rm -rf /tmp/mp_$n
    (cd /tmp; TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/mp_$n -o p -- python $R/tools/exp/msda_probe/probe.py $e > /dev/null 2>&1 < /dev/null)
    f=$(find /tmp/mp_$n -name '*counter_collection.csv' | head -1)
    echo "-- exp $e, $c" >> $O/r04_msda_lds_dma_probe.txt
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f 2>/dev/null < /dev/null | grep -A4 "msda_fwd" | head -6 >> $O/r04_msda_lds_dma_probe.txt
  done
done
timeout 300 python tools/exp/addmm_probe.py > $O/r04_addmm_probe.txt 2>&1 < /dev/null
timeout 300 python -m pytest tests/test_msda_gpu.py -x -q > $O/r04_msda_tests2.txt 2>&1 < /dev/null
tail -3 $O/r04_msda_tests2.txt; cat $O/r04_addmm_probe.txt; cat $O/r04_msda_lds_dma_probe.txt
