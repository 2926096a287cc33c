# per-clip kernel table of the timed steps (clip-by-clip mode: the marker separates steps)
R=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/steady; mkdir -p /tmp/steady
DVIS_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/steady -o kt -- python $R/bench.py --steps 5 --warmup 2 --clip-stream 0 --no-cpu-baseline --no-extra > /tmp/steady/run.log 2>&1
f=$(find /tmp/steady -name '*kernel_trace.csv' | head -1)
python $R/tools/steady_stats.py $f 60 > $R/gpurun_out/r03_steady_state_kernels.txt; grep "^{" /tmp/steady/run.log | cut -c1-160 >> $R/gpurun_out/r03_steady_state_kernels.txt
