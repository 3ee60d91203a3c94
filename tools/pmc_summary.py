"""Sum rocprofv3 --pmc counter CSVs per kernel name: python tools/pmc_summary.py <dir> [name filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt and flt not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, c in agg.items():
    print(k[:110])
    for n, v in sorted(c.items()):
        print(f"   {n:32s} {v / max(1, cnt[(k, n)]):16.1f} per dispatch ({cnt[(k, n)]} dispatches)")
