"""Mean of every counter per kernel name from a rocprofv3 --pmc output directory.  python tools/exp/pmc_summary.py DIR [substring]"""
import csv, glob, sys, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} mean={sum(v) / len(v):.4g}  n={len(v)}")
