#!/bin/bash
# Round-4 evidence run (one gpurun call): bench lines + steady-state kernel tables of configs #3, #2, #5 + tracker timeline.
R=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$R; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/r04_bench_line.json 2> $O/r04_bench.err < /dev/null
python bench.py --steps 10 --warmup 3 --mode online --frames 5 --no-cpu-baseline > $O/r04_bench_line_online_T5.json 2>> $O/r04_bench.err < /dev/null
python bench.py --steps 4 --warmup 2 --backbone vitl --queries 200 --no-cpu-baseline > $O/r04_bench_line_vitl_200q.json 2>> $O/r04_bench.err < /dev/null
bash tools/exp/steady.sh r04 < /dev/null
bash tools/exp/steady.sh r04_online_T5 --mode online --frames 5 < /dev/null
bash tools/exp/steady.sh r04_vitl_200q --backbone vitl --queries 200 < /dev/null
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/t2
DVIS_TT_ONLY_FUSED=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/t2 -o t2 -- python $R/tools/tracker_time.py 30 1 > $O/r04_tracker_time.txt 2>&1 < /dev/null
f=$(find /tmp/t2 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/timeline.py $f 76 100 > $O/r04_tracker_timeline.txt 2>&1 < /dev/null
cd $R; python tools/tracker_time.py 30 1 > $O/r04_tracker_time.txt 2>&1 < /dev/null
ls -la $O | tail -12
