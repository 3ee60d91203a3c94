"""Per-iteration kernel time table from a rocprofv3 *_kernel_stats.csv: python tools/prof_top.py <csv or dir> <iterations> [rows]"""
import csv, glob, os, sys
path, steps = sys.argv[1], int(sys.argv[2])
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{path}: {tot / 1e6 / steps:.2f} ms of kernels per iteration, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per iteration")
for r in rows[:nrows]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.2f} ms/it {float(r['TotalDurationNs']) / tot * 100:5.1f}% {int(r['Calls']) / steps:7.1f} calls/it {float(r['AverageNs']) / 1e3:8.1f} us  {n[:110]}")
