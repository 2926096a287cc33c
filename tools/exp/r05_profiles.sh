#!/bin/bash
# Round-5 closing profiles (run on the GPU box through gpurun): rocprofv3 kernel stats of the bench command, the steady-state
# per-clip kernel table, the memory-side PMC passes of the MSDeformAttn forward (-> r05_msda_traffic.json), PMC passes of the
# split-f16 kernels, the bench lines of every configuration, the per-rank emulation.
#   STAGE=1 bench lines + kernel stats + steady table;  STAGE=2 PMC passes;  STAGE=3 rank emulation;  default: all
mkdir -p gpurun_out/r05
S=${STAGE:-0}
if [ "$S" = "0" ] || [ "$S" = "1" ]; then
  python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r05/smoke.txt 2>&1
  python bench.py > gpurun_out/r05/r05_bench_line.json 2> gpurun_out/r05/r05_bench_line.err
  cut -c1-200 gpurun_out/r05/r05_bench_line.json
  PMC=0 bash tools/prof.sh r05_bench python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
  bash tools/exp/steady.sh r05 > /dev/null 2>&1
  python bench.py --mode online --frames 5 --steps 20 --warmup 4 --no-cpu-baseline --no-extra > gpurun_out/r05/r05_bench_line_online_T5.json 2>/dev/null
  python bench.py --frames 64 --steps 6 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r05/r05_bench_line_T64.json 2>/dev/null
  python bench.py --backbone vitl --queries 200 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/r05/r05_bench_line_vitl_200q.json 2>/dev/null
  python bench.py --task vis --no-cpu-baseline --no-extra > gpurun_out/r05/r05_bench_line_vis.json 2>/dev/null
  bash tools/exp/steady.sh r05_online_T5 --mode online --frames 5 > /dev/null 2>&1
  bash tools/exp/steady.sh r05_vitl_200q --backbone vitl --queries 200 > /dev/null 2>&1
  python tools/attn_time.py 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r05/r05_attn_time.txt
  python bench.py --clip-stream 0 --no-cpu-baseline --no-extra > gpurun_out/r05/r05_bench_line_clip_by_clip.json 2>/dev/null
  cp gpurun_out/prof/r05_bench_kernel_stats.csv gpurun_out/r05/r05_bench_kernel_stats.csv 2>/dev/null
  grep "^{" gpurun_out/prof/r05_bench_run.log | cut -c1-2000 > gpurun_out/r05/r05_bench_line_under_rocprof.json 2>/dev/null
  cp gpurun_out/r05_steady_state_kernels.txt gpurun_out/r05_online_T5_steady_state_kernels.txt gpurun_out/r05_vitl_200q_steady_state_kernels.txt gpurun_out/r05/ 2>/dev/null
fi
if [ "$S" = "0" ] || [ "$S" = "2" ]; then
  PMC_LIGHT=1 timeout 600 bash tools/prof.sh r05_pd python tools/pd_only.py pixel_decoder 5 > /dev/null 2>&1
  python tools/traffic_json.py gpurun_out/prof/r05_pd_pmc.txt gpurun_out/r05/r05_msda_traffic.json > /dev/null 2>&1
  cp gpurun_out/prof/r05_pd_pmc.txt gpurun_out/r05/r05_pd_pmc.txt 2>/dev/null
  (timeout 300 python tools/x3_time.py; timeout 200 python tools/x3_time.py conv; timeout 200 python tools/x3_time.py conv3) 2>&1 | grep -v "Warn\|amdgpu.ids\|return float" > gpurun_out/r05/r05_x3_time.txt
  PMC=1 timeout 500 bash tools/prof.sh r05x3_k python tools/x3_time.py 10 > /dev/null 2>&1
  PMC=1 timeout 500 bash tools/prof.sh r05x3_conv3 python tools/x3_time.py conv3 > /dev/null 2>&1
  cat gpurun_out/prof/r05x3_k_pmc.txt gpurun_out/prof/r05x3_conv3_pmc.txt > gpurun_out/r05/r05_x3_pmc_raw.txt 2>/dev/null
fi
if [ "$S" = "0" ] || [ "$S" = "3" ]; then
  timeout 900 python tools/rank_emulation.py 2>&1 | grep -v "Warn\|amdgpu.ids" > gpurun_out/r05/r05_rank_emulation.txt
fi
ls gpurun_out/r05 | tail -n 30
