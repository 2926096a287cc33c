#!/bin/bash
# Per-clip kernel table of the timed steps (clip-by-clip mode: the marker separates steps) for one bench configuration.
#   tools/exp/steady.sh <tag> [bench.py arguments...]      -> gpurun_out/<tag>_steady_state_kernels.txt
R=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$R
TAG=$1; shift
W=/tmp/steady_$TAG; rm -rf $W; mkdir -p $W $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
DVIS_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d $W -o kt -- python $R/bench.py --steps 5 --warmup 2 --clip-stream 0 --no-cpu-baseline --no-extra "$@" > $W/run.log 2>&1 < /dev/null
f=$(find $W -name '*kernel_trace.csv' | head -1)
OUT=$R/gpurun_out/${TAG}_steady_state_kernels.txt
if [ -z "$f" ]; then echo "no kernel trace (see run.log)" > $OUT; tail -5 $W/run.log >> $OUT; exit 0; fi
echo "# bench.py --steps 5 --warmup 2 --clip-stream 0 $@   (rocprofv3 --kernel-trace, timed steps only)" > $OUT
python $R/tools/steady_stats.py $f 60 >> $OUT 2>&1 < /dev/null
grep "^{" $W/run.log | cut -c1-300 >> $OUT
rm -rf $W
