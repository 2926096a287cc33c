#!/bin/bash
# Round-6 final checks on the GPU box: memset-under-capture repro, the RCCL single-rank tool (with the collective range-guard check),
# the multi-rank-on-one-GPU shard tests, then the closing bench line / kernel stats / steady tables at the final tree.
mkdir -p gpurun_out/r06
python tools/exp/memset_graph_repro.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tee gpurun_out/r06/memset_graph_repro.txt
(timeout 900 python -m pytest tests/test_shard_gpu.py tests/test_stream_gpu.py tests/test_x3_range_guard_gpu.py -x -q -m gpu 2>&1 | tail -5) | tee gpurun_out/r06/final_shard_tests.txt
STAGE=1 bash tools/exp/r06_profiles.sh 2>&1 | tail -22
