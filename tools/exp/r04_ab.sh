#!/bin/bash
# A/B on the GPU box: fused preprocess + one K/V compaction (in the build), conv1x1 dispatch order (DVIS_X3_CONV1X1_FIRST)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_fused_elementwise_gpu.py -q -x 2>&1 | tail -n 3
(timeout 200 python tools/x3_time.py conv) 2>&1 | grep -v "Warn\|amdgpu.ids\|return float" > gpurun_out/x3_conv_time2.txt
tail -n 6 gpurun_out/x3_conv_time2.txt | cut -c1-260
for f in 0 1 0 1; do
  DVIS_X3_CONV1X1_FIRST=$f python bench.py --no-cpu-baseline --no-extra 2>/dev/null > gpurun_out/ab_first$f.json
  echo "first=$f $(cut -c1-150 gpurun_out/ab_first$f.json)"
done
