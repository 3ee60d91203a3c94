"""Per-step kernel time by kernel family from a rocprofv3 *_kernel_stats.csv: python tools/prof_families.py <csv> <steps incl. warm-up> [rows]"""
import collections
import csv
import re
import sys

f, steps = sys.argv[1], float(sys.argv[2])
rows_n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(csv.DictReader(open(f)))
fam, calls = collections.Counter(), collections.Counter()
for r in rows:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([\w:]+)", n)
    k = m.group(1) if m else n[:30]
    if k == "igemm_f32":
        k += "<" + re.search(r"igemm_f32<(\w+)", n).group(1) + ">"
    if "at::native" in n:
        mm = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", n.split("kernel<", 1)[-1])
        k = "at::" + (mm.group(1) if mm else "other")
    fam[k] += float(r["TotalDurationNs"])
    calls[k] += int(r["Calls"])
tot = sum(fam.values())
print(f"== {f}: {tot / 1e6 / steps:.2f} ms of kernels per step, {sum(calls.values()) / steps:.0f} launches per step")
for k, v in fam.most_common(rows_n):
    print("%6.2f%%  %7.2f ms/step  x%5d/step  %s" % (100 * v / tot, v / 1e6 / steps, calls[k] / steps, k))
