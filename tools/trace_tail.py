"""The last part of one training iteration in a rocprofv3 kernel trace of bench.py, kernel by kernel:
python tools/trace_tail.py <run_kernel_trace.csv> [ms before the G-arena Adam launch to start at]  -> start offset, duration, gap to the previous
kernel on the same queue, queue, workgroups, name; then the window's summary per kernel family."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]),
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_k" in r[2]]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
i1 = adam[-1]
t1 = rows[i1][0]
seg = [r for r in rows[:i1 + 1] if r[0] >= t1 - win * 1e6]
last_end = {}
fam = {}
busy_events = []
for s, e, name, q, wgs in seg:
    n = re.sub(r"\(anonymous namespace\)::", "", name); n = re.sub(r"^void ", "", n)
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("%8.3f ms  %7.1f us  gap %6.1f  q%d  %6d wg  %s" % ((s - (t1 - win * 1e6)) / 1e6, (e - s) / 1e3, gap, q, wgs, n[:95]))
    f = n.split("<")[0].split("(")[0]
    fam[f] = fam.get(f, [0, 0.0]); fam[f][0] += 1; fam[f][1] += (e - s) / 1e6
    busy_events += [(s, 1), (e, -1)]
busy_events.sort(); cur = 0; last = seg[0][0]; busy = 0
for t, d in busy_events:
    if cur > 0: busy += t - last
    cur += d; last = t
print("window %.1f ms: >= 1 kernel running %.2f ms, %d launches" % (win, busy / 1e6, len(seg)))
for k, (c, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %7.2f ms %4d  %s" % (ms, c, k))
