R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
DVIS_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/ss -o kt -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --clip-stream 0 --no-extra > /tmp/ss.log 2>&1
f=$(find /tmp/ss -name '*kernel_trace.csv' | head -1)
mkdir -p $R/gpurun_out/prof
KNAME=110 python $R/tools/steady_stats.py $f 60 > $R/gpurun_out/prof/steady_now.txt
head -64 $R/gpurun_out/prof/steady_now.txt | cut -c1-170
