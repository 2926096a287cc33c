export DVIS_BENCH_WATCHDOG=250
for n in 2 3; do
 timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n tools/stream_shard_check.py --clips 4 --frames 6 2>&1 | grep -E "^clip |SHARD_CHECK|rror|^rank " | head -12
done
DVIS_DIST_BACKEND=nccl DVIS_FORCE_COLLECTIVES=1 timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29539 tools/stream_shard_check.py --clips 3 --frames 6 2>&1 | grep -E "^clip |SHARD_CHECK|rror|^rank " | head -12
for n in 2 4; do
 DVIS_BENCH_ONE_DEVICE=1 DVIS_DIST_BACKEND=gloo timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n --steps 8 --warmup 1 --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
done
python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
