#!/bin/bash
# Round-6 (second half) closing profiles (run on the GPU box through gpurun): the default bench line (carries configs #2 / #4-length / #5 and every
# roofline entry), rocprofv3 kernel stats of the bench command, steady-state per-clip kernel tables (R50 headline, ViT-L), the memory-side
# PMC passes of the MSDeformAttn forward (-> r06b_msda_traffic.json) and of the split-f16 kernel families over the BENCH command
# (-> r06b_x3_traffic.json: what bench.py's roofline_conv_x3 / roofline_ffn `traffic` reads).
#   STAGE=1 bench line + kernel stats + steady tables;  STAGE=2 PMC passes;  default: both
mkdir -p gpurun_out/r06b
S=${STAGE:-0}
if [ "$S" = "0" ] || [ "$S" = "2" ]; then
  PMC_LIGHT=1 timeout 600 bash tools/prof.sh r06b_pd python tools/pd_only.py pixel_decoder 5 > /dev/null 2>&1
  python tools/traffic_json.py gpurun_out/prof/r06b_pd_pmc.txt gpurun_out/r06b/r06b_msda_traffic.json > /dev/null 2>&1
  cp gpurun_out/prof/r06b_pd_pmc.txt gpurun_out/r06b/r06b_pd_pmc.txt 2>/dev/null
  PMC_LIGHT=1 timeout 900 bash tools/prof.sh r06b_benchpmc python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
  python tools/x3_traffic_json.py gpurun_out/prof/r06b_benchpmc_pmc.txt gpurun_out/r06b/r06b_x3_traffic.json > /dev/null 2>&1
  grep -E "^(conv1x1_x3|x3_|msda_fwd|mask_gemm|attn_keysplit|bneck_chain|upsample_add_image)" -A4 gpurun_out/prof/r06b_benchpmc_pmc.txt > gpurun_out/r06b/r06b_bench_pmc_own_kernels.txt 2>/dev/null
  # the traffic files must be in place BEFORE the bench line is taken (bench.py reads them from profiles/)
  cp gpurun_out/r06b/r06b_msda_traffic.json gpurun_out/r06b/r06b_x3_traffic.json profiles/ 2>/dev/null
fi
if [ "$S" = "0" ] || [ "$S" = "1" ]; then
  python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r06b/smoke.txt 2>&1
  (time python bench.py) > gpurun_out/r06b/r06b_bench_line.json 2> gpurun_out/r06b/r06b_bench_line.err
  cut -c1-200 gpurun_out/r06b/r06b_bench_line.json
  PMC=0 bash tools/prof.sh r06b_bench python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
  cp gpurun_out/prof/r06b_bench_kernel_stats.csv gpurun_out/r06b/r06b_bench_kernel_stats.csv 2>/dev/null
  grep "^{" gpurun_out/prof/r06b_bench_run.log | cut -c1-3000 > gpurun_out/r06b/r06b_bench_line_under_rocprof.json 2>/dev/null
  bash tools/exp/steady.sh r06b > /dev/null 2>&1
  bash tools/exp/steady.sh r06b_vitl_200q --backbone vitl --queries 200 > /dev/null 2>&1
  bash tools/exp/steady.sh r06b_online_T5 --mode online --frames 5 > /dev/null 2>&1
  cp gpurun_out/r06b_steady_state_kernels.txt gpurun_out/r06b_vitl_200q_steady_state_kernels.txt gpurun_out/r06b_online_T5_steady_state_kernels.txt gpurun_out/r06b/ 2>/dev/null
fi
ls gpurun_out/r06b | tail -n 40
