"""Per-clip kernel time of the TIMED steps of bench.py from a rocprofv3 --kernel-trace CSV.

bench.py (env DVIS_BENCH_MARK=1) launches a marker kernel (torch's uniform_ RNG kernel, used nowhere else) at the
start of every timed step; everything before the first marker (warm-up, MIOpen find, calibration) is dropped.
    python tools/steady_stats.py kernel_trace.csv [topN]
"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
s_k = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Start"
e_k = "End_Timestamp" if "End_Timestamp" in rows[0] else "End"
rows.sort(key=lambda r: int(r[s_k]))
marks = [i for i, r in enumerate(rows) if "distribution_elementwise" in r[name_k] or "uniform" in r[name_k].lower()]
if not marks:
    sys.exit("no marker kernels found (run bench.py with DVIS_BENCH_MARK=1)")
steps = len(marks)
sel = rows[marks[0]:]
t0, t1 = int(sel[0][s_k]), max(int(r[e_k]) for r in sel)
agg = defaultdict(lambda: [0, 0])
for r in sel:
    n = r[name_k].replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)[:int(__import__("os").environ.get("KNAME", "100"))]
    agg[n][0] += int(r[e_k]) - int(r[s_k])
    agg[n][1] += 1
busy = sum(v[0] for v in agg.values())
print(f"{steps} timed steps, wall {(t1 - t0) / 1e6 / steps:.2f} ms/clip, kernel time {busy / 1e6 / steps:.2f} ms/clip, "
      f"{sum(v[1] for v in agg.values()) / steps:.0f} launches/clip")
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{d / 1e6 / steps:8.3f} ms/clip {100.0 * d / busy:5.1f}%  {c / steps:7.1f} calls/clip  avg {d / c / 1e3:8.1f} us  {n}")
