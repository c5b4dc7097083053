"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files:  python tools/pmc_agg.py DIR [name filter]"""
import collections, csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k[:110])
    for c in sorted(agg[k]):
        n, v = agg[k][c]
        print(f"    {c:42s} {v / n:18.1f} per launch ({n} launches)")
