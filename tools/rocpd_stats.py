"""Kernel-stats CSV (same columns as rocprofv3 --stats' *_kernel_stats.csv) from a rocprofv3 rocpd database:
python tools/rocpd_stats.py <run_results.db> <out.csv>"""
import csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
dur = "duration" if "duration" in cols else '("end" - start)'
rows = con.execute(f"select name, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, lo, hi in rows:
        w.writerow([n, c, t, f"{a:.1f}", f"{100.0 * t / tot:.2f}", lo, hi])
print(f"{out}: {len(rows)} kernels, {tot / 1e6:.1f} ms")
