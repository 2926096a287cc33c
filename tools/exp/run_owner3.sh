export DVIS_BENCH_WATCHDOG=250
for i in 1 2; do
timeout 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 2953$i tools/stream_shard_check.py --clips 4 --frames 6 2>&1 | grep -E "clip |SHARD_CHECK|rror|rank " | head -40
done
