#!/bin/bash
# Round-4 probes (one gpurun call): MSDA backward old / new, MSDA slot orders (time + PMC), TA -> LDS corner probe, rank emulation.
R=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$R; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py -x -q > $O/r04_msda_tests.txt 2>&1 < /dev/null
timeout 300 python tools/msda_bwd_time.py > $O/r04_msda_bwd_time.txt 2>&1 < /dev/null
# (a) slot orders: 0 = reference row order, 1 = [offsets | logits | pad], 2 = [logits | offsets | pad]
: > $O/r04_msda_slot_orders.txt
for s in 0 1 2; do
  echo "== DVIS_MSDA_SLOTS=$s" >> $O/r04_msda_slot_orders.txt
  DVIS_MSDA_SLOTS=$s timeout 300 python tools/msda_real.py 2>&1 < /dev/null | grep -v ids >> $O/r04_msda_slot_orders.txt
  for c in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d" " -f1); rm -rf /tmp/mt_$n
    (cd /tmp; TMPDIR=/tmp DVIS_MSDA_SLOTS=$s timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/mt_$n -o p -- python $R/tools/msda_real.py > /dev/null 2>&1 < /dev/null)
    f=$(find /tmp/mt_$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f 2>/dev/null < /dev/null | grep -A4 "^msda_fwd" | head -6 >> $O/r04_msda_slot_orders.txt
  done
done
# (b) corner lines of the coarsest level TA -> LDS -> ds_read instead of TA -> VGPR
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
timeout 900 python tools/rank_emulation.py --worlds 1,2,4,8 --clips 16 > $O/r04_rank_emulation.txt 2>&1 < /dev/null
ls -la $O | tail -8
