# per-clip kernel table of the timed steps (clip-by-clip mode: the marker separates steps)
R=${GRAFT_REPO_ROOT:-$PWD}; export PYTHONPATH=$R; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/steady; mkdir -p /tmp/steady
DVIS_BENCH_MARK=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/steady -o kt -- python $R/bench.py --steps 5 --warmup 2 --clip-stream 0 --no-cpu-baseline --no-extra > /tmp/steady/run.log 2>&1
f=$(find /tmp/steady -name '*kernel_trace.csv' | head -1)
python $R/tools/steady_stats.py $f 60 > $R/gpurun_out/r03_steady_state_kernels.txt; grep "^{" /tmp/steady/run.log | cut -c1-160 >> $R/gpurun_out/r03_steady_state_kernels.txt
# the torch element-wise kernels of the timed steps, largest first (what is left to fuse)
python - "$f" >> $R/gpurun_out/r03_steady_state_kernels.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
s_k = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Start"
e_k = "End_Timestamp" if "End_Timestamp" in rows[0] else "End"
rows.sort(key=lambda r: int(r[s_k]))
marks = [i for i, r in enumerate(rows) if "distribution_elementwise" in r[name_k]]
sel = rows[marks[0]:] if marks else rows
steps = max(1, len(marks))
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    n = r[name_k]
    if "at::native" not in n:
        continue
    tag = "add<float>" if "CUDAFunctor_add<float>" in n else ("copy" if "direct_copy" in n else n[18:70])
    g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    agg[(tag, g)][0] += int(r[e_k]) - int(r[s_k]); agg[(tag, g)][1] += 1
print("\ntorch element-wise kernels by (kind, grid size), per clip:")
for (tag, g), (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {d / 1e6 / steps:7.3f} ms/clip  {c / steps:6.1f} calls/clip  avg {d / c / 1e3:7.1f} us  grid {g:>10}  {tag}")
PY
