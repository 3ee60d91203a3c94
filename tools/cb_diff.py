"""Side-by-side of two tools/conv_bench.py outputs: python tools/cb_diff.py a.txt b.txt"""
import sys, re
def rd(p):
    d = {}
    for l in open(p):
        m = re.match(r"(.{36}) +([\d.]+) \| +([\d.]+) +([\d.]+) \| +([\d.]+) +([\d.]+) \| +([\d.]+) +([\d.]+)", l)
        if m: d[m.group(1).strip()] = [float(m.group(i)) for i in (3, 5, 7)]
    return d
a, b = rd(sys.argv[1]), rd(sys.argv[2])
ta = tb = 0
print(f"{'shape':36s} | {'fwd a':>7s} {'b':>7s} {'%':>5s} | {'bwdD a':>7s} {'b':>7s} {'%':>5s} | {'bwdW a':>7s} {'b':>7s} {'%':>5s}")
for k in a:
    if k not in b: continue
    row = f"{k:36s}"
    for i in range(3):
        row += f" | {a[k][i]:7.3f} {b[k][i]:7.3f} {100 * (b[k][i] / a[k][i] - 1):+5.0f}"
        ta += a[k][i]; tb += b[k][i]
    print(row)
print(f"total {ta:.2f} -> {tb:.2f} ms ({100 * (tb / ta - 1):+.1f} %)")
