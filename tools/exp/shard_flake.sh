#!/bin/bash
# Development: how often does tools/stream_shard_check.py (N ranks on one GPU) fail, and with what output?
#   bash tools/exp/shard_flake.sh <tag> <runs> [world] [frames] [ENV=VALUE ...]
tag=$1; n=$2; world=${3:-3}; frames=${4:-4}; shift; shift; [ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
mkdir -p gpurun_out
ok=0; bad=0
for i in $(seq 1 $n); do
  env "$@" MASTER_ADDR=127.0.0.1 DVIS_DIST_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$world --master-addr 127.0.0.1 --master-port $((29600 + i)) tools/stream_shard_check.py --clips 3 --frames $frames --out /tmp/sc_$tag$i > /tmp/o_$tag$i.txt 2>&1
  if grep -aq "SHARD_CHECK OK" /tmp/o_$tag$i.txt; then ok=$((ok+1)); else bad=$((bad+1)); grep -av "Warning\|warn\|amdgpu.ids\|socket.cpp\|^$" /tmp/o_$tag$i.txt | cut -c1-400 > gpurun_out/shard_fail_$tag$i.txt; fi
done
echo "== $tag (world $world, $frames frames): ok $ok bad $bad"
