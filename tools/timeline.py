"""Kernel-by-kernel timeline from a rocprofv3 kernel_trace.csv: the LAST `count` kernels before the final `skip` ones —
start offset, duration, gap to the previous kernel's end, grid, short name.  For launch-bound chains (the tracker's hipGraph).
    python tools/timeline.py <kernel_trace.csv> [count] [skip]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
count = int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 200
nk = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
sk = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Start"
ek = "End_Timestamp" if "End_Timestamp" in rows[0] else "End"
rows.sort(key=lambda r: int(r[sk]))
sel = rows[len(rows) - skip - count:len(rows) - skip]
t0 = int(sel[0][sk])
prev_end = None
tot_d = tot_g = 0
for r in sel:
    s, e = int(r[sk]), int(r[ek])
    name = re.sub(r"\(anonymous namespace\)::", "", r[nk])
    name = re.sub(r"^void ", "", name).split("(")[0][:70]
    gap = 0 if prev_end is None else s - prev_end
    g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
    w = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  grid {g:>8}/{w:<4} {name}")
    prev_end = e
    tot_d += e - s
    tot_g += max(gap, 0)
print(f"sum of durations {tot_d / 1e3:.1f} us, sum of gaps {tot_g / 1e3:.1f} us over {len(sel)} kernels")
