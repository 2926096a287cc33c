"""Compact view of a rocprofv3 kernel_stats.csv: calls, average / min us, share, short kernel name.
    python tools/kstats_short.py <kernel_stats.csv> [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0][:90]
    print(f"{int(r['Calls']):7d} calls  avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  "
          f"{float(r['Percentage']):5.1f} %  {name}")
