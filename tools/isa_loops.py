"""Instruction mix of the loops of one kernel in a hipcc -S listing: python tools/isa_loops.py file.s <substring of the mangled name>."""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and pat in l.split(":")[0])
end = next(j for j in range(start, len(src)) if src[j].startswith(".Lfunc_end"))
body = src[start:end]
print(src[start].split(":")[0], len(body), "lines")
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and labels.get(m.group(1), 1 << 30) < i:
        a = labels[m.group(1)]
        cnt = collections.Counter()
        for s in body[a:i]:
            s = s.strip()
            if not s or s[0] in ";." :
                continue
            cnt[s.split()[0]] += 1
        tot = sum(cnt.values())
        mf = sum(v for k, v in cnt.items() if "mfma" in k)
        va = sum(v for k, v in cnt.items() if k.startswith("v_") and "mfma" not in k)
        print("loop", m.group(1), "lines", a, i, "instrs", tot, "mfma", mf, "valu", va)
        if mf > 20 or (len(sys.argv) > 3):
            for k, v in cnt.most_common(45):
                print("    %-28s %d" % (k, v))
for l in src[end:end + 60]:
    if any(k in l for k in ("vgpr_count", "sgpr_count", "lds_size", "spill", "Occupancy", "NumVgprs", "NumAgprs", "ScratchSize")):
        print(l.strip())
