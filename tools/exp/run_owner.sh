# development check on ONE box: several ranks share cuda:0, collectives through gloo (never used for reported numbers)
export DVIS_BENCH_WATCHDOG=250
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 tools/stream_shard_check.py --clips 3 --frames 6 2>&1 | grep -E "^clip |SHARD_CHECK|rror|^rank " | head -8
DVIS_BENCH_ONE_DEVICE=1 DVIS_DIST_BACKEND=gloo timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep metric | cut -c1-130,400-640
DVIS_DIST_BACKEND=nccl DVIS_FORCE_COLLECTIVES=1 timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep metric | cut -c1-130
