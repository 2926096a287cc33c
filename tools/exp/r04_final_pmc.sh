#!/bin/bash
# Round-4 closing pass on the GPU box: the bench line with roofline_conv_x3, which aten ops still launch kernels, timings + PMC
# passes of the split-f16 kernels at the final commit.
mkdir -p gpurun_out
python bench.py > gpurun_out/r04_bench_line_b.json 2> gpurun_out/r04_bench_line_b.err
cut -c1-200 gpurun_out/r04_bench_line_b.json
timeout 200 python tools/exp/torch_ops_hunt.py all 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r04_torch_ops_hunt.txt
head -n 30 gpurun_out/r04_torch_ops_hunt.txt | cut -c1-220
(timeout 300 python tools/x3_time.py; timeout 200 python tools/x3_time.py conv; timeout 200 python tools/x3_time.py conv3) 2>&1 | grep -v "Warn\|amdgpu.ids\|return float" > gpurun_out/r04_x3_time.txt
PMC=1 timeout 400 bash tools/prof.sh r04x3_k python tools/x3_time.py 10 > /dev/null 2>&1
PMC=1 timeout 400 bash tools/prof.sh r04x3_conv3 python tools/x3_time.py conv3 > /dev/null 2>&1
ls gpurun_out/prof | tail -n 12
