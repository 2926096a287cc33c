#!/bin/bash
# rocprofv3 helper for the GPU box (run via gpurun):  tools/prof.sh <tag> <cmd...>
#   pass 1: --kernel-trace --stats        -> gpurun_out/prof/<tag>_kernel_stats.csv (+ top of it in <tag>_stats.txt)
#   pass 2..: one --pmc pass per counter group (never combined with other trace domains) -> <tag>_pmc.txt
# Only small summaries are kept (gpurun_out/ is capped at 64 MiB); copy what should be judged into profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
# rocprofv3 wants cwd=/tmp, TMPDIR=/tmp on this pool: make repo-relative arguments absolute first
ARGS=()
for a in "$@"; do if [ -e "$R/$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done
set -- "${ARGS[@]}"
export PYTHONPATH=$R:${PYTHONPATH:-}
cd /tmp && export TMPDIR=/tmp
W=/tmp/prof_$TAG; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- "$@" > $OUT/${TAG}_run.log 2>&1
f=$(find $W/kt -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats.csv && head -25 "$f" > $OUT/${TAG}_stats.txt
if [ "${PMC:-1}" = "1" ]; then
  : > $OUT/${TAG}_pmc.txt
  # PMC_LIGHT=1: only the four memory-side groups tools/traffic_json.py needs (HBM bytes, L2 hit rate, TA busy)
  if [ "${PMC_LIGHT:-0}" = "1" ]; then
    GROUPS_=("TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")
  else
    GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum")
  fi
  for grp in "${GROUPS_[@]}"; do
    n=$(echo $grp | cut -d" " -f1)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $W/pmc_$n -o pmc -- "$@" > $W/pmc_$n.log 2>&1
    c=$(find $W/pmc_$n -name '*counter_collection.csv' | head -1)
    if [ -n "$c" ]; then python3 $R/tools/pmc_summary.py "$c" >> $OUT/${TAG}_pmc.txt; else echo "no counters for: $grp" >> $OUT/${TAG}_pmc.txt; tail -3 $W/pmc_$n.log >> $OUT/${TAG}_pmc.txt; fi
  done
fi
rm -rf $W
ls -la $OUT | tail -8
