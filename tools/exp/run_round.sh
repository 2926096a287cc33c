export DVIS_BENCH_WATCHDOG=600
for n in 2 3; do
DVIS_BENCH_ONE_DEVICE=1 DVIS_DIST_BACKEND=gloo timeout 580 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --steps 7 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-130,400-640
done
python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-130,400-640
