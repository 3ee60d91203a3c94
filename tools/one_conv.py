"""Run one convolution shape a few times (for rocprofv3 --pmc / --kernel-trace): one_conv.py B Cin H Cout ks stride pad [pass]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
from agl import lib as L
B, Cin, H, Cout, ks, s, p = (int(v) for v in sys.argv[1:8])
which = sys.argv[8] if len(sys.argv) > 8 else "fwd"
if os.environ.get("AGL_PREC"): L.set_conv_precision(os.environ["AGL_PREC"])
if os.environ.get("AGL_SPLIT3"): L.CONV_FLAGS |= L.CONV_SPLIT3
if os.environ.get("AGL_W8"): L.CONV_FLAGS |= L.CONV_W8
if os.environ.get("AGL_PRIO"): L.CONV_FLAGS |= int(os.environ["AGL_PRIO"]) << 14
if os.environ.get("AGL_ABLATE"): L.CONV_FLAGS |= int(os.environ["AGL_ABLATE"]) << 9
x = torch.randn(B, Cin, H, H, device="cuda:0"); w = torch.randn(Cout, Cin, ks, ks, device="cuda:0") * 0.05
y = L.conv2d_fwd(x, w, None, s, p); dy = torch.randn_like(y)
if os.environ.get("AGL_BLOCKED"):      # channel-blocked bf16 x (and y for 3x3 stride 1): the forms of the discriminators' block chain
    xb = L.to_blocked_dev(x)
    wsrc = L.WeightSrc(torch.nn.Parameter(w), lambda: 0)
    fwd = lambda: L.conv2d_fwd(xb, w, None, s, p, in_relu=True, relu=(ks == 3), out_blk=(ks == 3 and s == 1), wsrc=wsrc)
    for _ in range(25): fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fwd()
    e1.record(); torch.cuda.synchronize()
    print("%s blocked fwd: %.1f us per call" % (sys.argv[1:8], 1e3 * e0.elapsed_time(e1) / 20))
    sys.exit(0)
for _ in range(5):
    if which == "fwd": L.conv2d_fwd(x, w, None, s, p)
    elif which == "bwd_data": L.conv2d_bwd_data(dy, w, (H, H), s, p)
    else: L.conv2d_bwd_weight(dy, x, ks, s, p)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    if which == "fwd": L.conv2d_fwd(x, w, None, s, p)
    elif which == "bwd_data": L.conv2d_bwd_data(dy, w, (H, H), s, p)
    else: L.conv2d_bwd_weight(dy, x, ks, s, p)
e1.record(); torch.cuda.synchronize()
print("%s %s stagger=%s ablate=%s: %.1f us per call" % (sys.argv[1:8], which, os.environ.get("AGL_PRIO", "0"), os.environ.get("AGL_ABLATE", "0"), 1e3 * e0.elapsed_time(e1) / 20))
