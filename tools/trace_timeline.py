"""Timeline statistics of a rocprofv3 kernel trace of bench.py: python tools/trace_timeline.py <run_kernel_trace.csv> <steps in the trace>
Splits the trace into steps at the Adam kernel pairs, and for the last step prints: wall time, time with >= 1 kernel running,
idle gaps, sum of kernel durations (average concurrency), and per-phase (between Adam launches) numbers."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]),
                 int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_k" in r[2]]
print(len(rows), "dispatches,", len(adam), "adam launches")
# a step = two adam launches (D arena, G arena); take the span between the G-adam of step n-2 and the G-adam of step n-1
if len(adam) >= 4:
    i0, i1 = adam[-3], adam[-1]
else:
    i0, i1 = 0, len(rows) - 1
seg = rows[i0 + 1:i1 + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
wall = (t1 - t0) / 1e6
events = []
for s, e, *_ in seg:
    events.append((s, 1)); events.append((e, -1))
events.sort()
busy = 0; cur = 0; last = t0; conc_time = {}
for t, d in events:
    if cur > 0:
        busy += t - last
    conc_time[cur] = conc_time.get(cur, 0) + (t - last)
    cur += d; last = t
tot = sum(e - s for s, e, *_ in seg) / 1e6
print(f"step: wall {wall:.2f} ms, >=1 kernel running {busy / 1e6:.2f} ms ({100 * busy / 1e6 / wall:.1f} %), idle {wall - busy / 1e6:.2f} ms, "
      f"sum of kernel durations {tot:.2f} ms (avg concurrency {tot / wall:.2f}), {len(seg)} launches")
print("time by number of kernels in flight:", {k: round(v / 1e6, 2) for k, v in sorted(conc_time.items())})
# small-grid time: kernels with < 256 workgroups running alone
small_alone = 0
for k, (s, e, name, q, wgs) in enumerate(seg):
    pass
# per-queue busy
qs = {}
for s, e, name, q, wgs in seg:
    qs[q] = qs.get(q, 0) + (e - s)
print("kernel time per HW queue (ms):", {q: round(v / 1e6, 1) for q, v in sorted(qs.items())})
# biggest idle gaps
gaps = []
cur = 0; last = t0
for t, d in events:
    if cur == 0 and t - last > 5000:
        gaps.append((t - last, last - t0))
    cur += d; last = t
gaps.sort(reverse=True)
print("largest idle gaps (us @ ms into the step):", [(round(g / 1e3, 1), round(at / 1e6, 1)) for g, at in gaps[:12]])
print("idle in gaps > 5 us:", round(sum(g for g, _ in gaps) / 1e6, 2), "ms in", len(gaps), "gaps")
