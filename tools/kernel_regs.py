"""Per-kernel resource table of a `hipcc --cuda-device-only -S` listing: python tools/kernel_regs.py file.s [substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    rows.append((name, g("next_free_vgpr"), g("accum_offset"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, v, a, l, s), d in zip(rows, names):
    d = d.replace("(anonymous namespace)::", "").replace("void ", "")
    if pat in d:
        print("%-90s vgpr+agpr %4s accum_offset %4s lds %6s scratch %5s" % (d[:90], v, a, l, s))
