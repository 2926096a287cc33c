"""Print the top of a rocprofv3 kernel_stats CSV in a compact form:  python tools/kstats.py file.csv [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['Percentage']):6.2f}% calls={r['Calls']:>6} "
          f"avg_us={float(r['AverageNs']) / 1e3:9.1f}  {name[:120]}")
