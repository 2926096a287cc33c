export DVIS_BENCH_WATCHDOG=600
python bench.py --no-cpu-baseline --mode online --frames 5 2>&1 | grep metric | cut -c1-110
python bench.py --no-cpu-baseline --frames 64 --steps 5 2>&1 | grep metric | cut -c1-110
python bench.py --no-cpu-baseline --task vis 2>&1 | grep metric | cut -c1-110
python bench.py --no-cpu-baseline --task vss 2>&1 | grep metric | cut -c1-110
python bench.py --no-cpu-baseline --backbone vitl --queries 200 --steps 4 --warmup 1 2>&1 | grep metric | cut -c1-110
python bench.py --no-cpu-baseline --clip-stream 0 2>&1 | grep metric | cut -c1-110
