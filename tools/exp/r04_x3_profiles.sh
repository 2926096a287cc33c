#!/bin/bash
# Round-4 profiles of the split-f16 build (run on the GPU box through gpurun): rocprofv3 kernel stats of the bench command, the
# steady-state per-clip kernel table, timings + PMC passes of the x3 kernels, the bench lines of the configurations.
mkdir -p gpurun_out
PMC=0 bash tools/prof.sh r04x3_bench python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
bash tools/exp/steady.sh r04x3 > /dev/null 2>&1
if [ "${X3_PMC:-0}" = "1" ]; then
  (timeout 300 python tools/x3_time.py; timeout 200 python tools/x3_time.py conv; timeout 200 python tools/x3_time.py conv3) 2>&1 | grep -v "Warn\|amdgpu.ids\|return float" > gpurun_out/r04_x3_time.txt
  PMC=1 bash tools/prof.sh r04x3_k python tools/x3_time.py 10 > /dev/null 2>&1
  PMC=1 bash tools/prof.sh r04x3_conv3 python tools/x3_time.py conv3 > /dev/null 2>&1
fi
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.txt 2>&1
python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
python bench.py --mode online --frames 5 --no-cpu-baseline --no-extra > gpurun_out/r04_bench_line_online_T5.json 2>/dev/null
python bench.py --frames 64 --steps 6 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r04_bench_line_T64.json 2>/dev/null
python bench.py --clip-stream 0 --no-cpu-baseline --no-extra > gpurun_out/r04_bench_line_clip_by_clip.json 2>/dev/null
python bench.py --backbone vitl --queries 200 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r04_bench_line_vitl_200q_x3.json 2>/dev/null
tail -n 2 gpurun_out/smoke.txt; cut -c1-160 gpurun_out/r04_bench_line.json
