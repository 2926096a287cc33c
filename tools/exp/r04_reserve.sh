#!/bin/bash
# CUs left to phase B under stream() at the round's final balance of the two phases (phase A is 30 % shorter than when 32 was chosen)
mkdir -p gpurun_out
(timeout 120 python tools/x3_time.py; timeout 100 python tools/x3_time.py conv; timeout 100 python tools/x3_time.py conv3) 2>&1 | grep -v "Warn\|amdgpu.ids\|return float\|Consider\|Triggered" > gpurun_out/r04_x3_time.txt
: > gpurun_out/r04_reserve_sweep.txt
for r in 32 16 48 32 24 40; do
  DVIS_X3_RESERVE=$r python bench.py --no-cpu-baseline --no-extra 2>/dev/null > gpurun_out/reserve_$r.json
  echo "DVIS_X3_RESERVE=$r $(python -c "import json;d=json.load(open('gpurun_out/reserve_$r.json'));print(d['value'], d['ms_per_step'])")" | tee -a gpurun_out/r04_reserve_sweep.txt
done
