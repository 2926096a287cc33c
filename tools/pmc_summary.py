"""Summarise a rocprofv3 counter_collection CSV: per kernel name, dispatch count and mean of each counter."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    print("empty", sys.argv[1])
    sys.exit(0)
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "?")
    m = re.search(r"([A-Za-z_][A-Za-z0-9_]*(?:<[^(]*>)?)\(", name.replace("(anonymous namespace)::", ""))
    short = (m.group(1) if m else name)[:90]
    agg[short][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0) or 0))
for k, cs in sorted(agg.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
    n = max(len(v) for v in cs.values())
    print(f"{k}  dispatches={n}")
    for c, v in sorted(cs.items()):
        print(f"    {c:36s} mean={sum(v) / len(v):.4g}  sum={sum(v):.4g}")
