export DVIS_BENCH_WATCHDOG=150
python bench.py --no-cpu-baseline --frames 64 --steps 3 --warmup 1 2>&1 | grep -E "metric|Timeout|File" | cut -c1-150 | head -4
