"""Which BatchNorm statistic calls of one training iteration are served by the producing convolution (GPU box only)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
sys.path.insert(0, ROOT)
import torch
from agl import functional as F, lib as L
import bench

log = collections.Counter()
orig_bn, orig_fp, orig_cs, orig_c = L.bn_stats, L.bn_stats_from_partials, L.conv2d_fwd_stats, L.conv2d_fwd
last = [None]
def conv_stats(x, w, *k, **kw):
    y, part, rows = orig_cs(x, w, *k, **kw)
    last[0] = ("conv_stats", tuple(x.shape), tuple(w.shape), k, rows)
    return y, part, rows
def conv(x, w, *k, **kw):
    last[0] = ("conv", tuple(x.shape), tuple(w.shape), k, kw.get("accumulate", False))
    return orig_c(x, w, *k, **kw)
def bn(x, *k, **kw):
    log[("unfused", tuple(x.shape), str(last[0]))] += 1
    return orig_bn(x, *k, **kw)
def fp(part, rows, C, cnt, *k, **kw):
    log[("FUSED", C, cnt)] += 1
    return orig_fp(part, rows, C, cnt, *k, **kw)
L.bn_stats, L.bn_stats_from_partials, L.conv2d_fwd_stats, L.conv2d_fwd = bn, fp, conv_stats, conv
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--no-roofline"]
bench.main()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
