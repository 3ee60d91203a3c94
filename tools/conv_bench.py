"""Per-shape timing of the convolution kernels on the shapes of the 64 px / batch-64 train step (GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
from agl import lib as L

dev = "cuda:0"
if os.environ.get("AGL_NOPATCH"):
    L.CONV_FLAGS |= L.CONV_NO_PATCH
if os.environ.get("AGL_NOPATCH_S2"):
    L.CONV_FLAGS |= L.CONV_NO_PATCH_S2
if os.environ.get("AGL_POS_ALL"):
    L.CONV_FLAGS |= L.CONV_POS_ALL_KS
if os.environ.get("AGL_NOPOS"):
    L.CONV_FLAGS |= L.CONV_NO_POS
if os.environ.get("AGL_ANYGRID"):
    L.CONV_FLAGS |= L.CONV_ANY_GRID
if os.environ.get("AGL_SPLIT3"):
    L.CONV_FLAGS |= L.CONV_SPLIT3
if os.environ.get("AGL_W8"):
    L.CONV_FLAGS |= L.CONV_W8
if os.environ.get("AGL_PRIO"):
    L.CONV_FLAGS |= L.CONV_PRIO
if os.environ.get("AGL_PREC"):
    L.set_conv_precision(os.environ["AGL_PREC"])
O, N = 393, 64
# name, batch, Cin, H, Cout, ks, stride, pad
SHAPES = [
    ("LE.c2 64>128 k4s2 @66", O, 64, 66, 128, 4, 2, 1),
    ("LE.c3 128>256 k4s2 @33", O, 128, 33, 256, 4, 2, 1),
    ("LE.c4 256>512 k4s2 @16", O, 256, 16, 512, 4, 2, 1),
    ("CLSTM0.x 512>512 k5 @8", O, 512, 8, 512, 5, 1, 2),
    ("CLSTM0.h 128>512 k5 @8 (B=64)", N, 128, 8, 512, 5, 1, 2),
    ("CLSTM0.h 128>512 k5 @8 (B=48)", 48, 128, 8, 512, 5, 1, 2),
    ("CLSTM0.h 128>512 k5 @8 (B=32)", 32, 128, 8, 512, 5, 1, 2),
    ("CLSTM0.h 128>512 k5 @8 (B=16)", 16, 128, 8, 512, 5, 1, 2),
    ("CLSTM0.h 128>512 k5 @8 (B=8)", 8, 128, 8, 512, 5, 1, 2),
    ("CLSTM1.h 64>256 k5 @8 (B=32)", 32, 64, 8, 256, 5, 1, 2),
    ("RES 64>64 k3 @8 (B=64)", 64, 64, 8, 64, 3, 1, 1),
    ("CLSTM1.x 128>256 k5 @8", O, 128, 8, 256, 5, 1, 2),
    ("CLSTM1.h 64>256 k5 @8 (B=64)", N, 64, 8, 256, 5, 1, 2),
    ("CE.c1 3>64 k7 @32", O, 3, 32, 64, 7, 1, 3),
    ("CE.c2 64>128 k4s2 @32", O, 64, 32, 128, 4, 2, 1),
    ("CE.c3 128>256 k4s2 @16", O, 128, 16, 256, 4, 2, 1),
    ("CE.c4 256>512 k4s2 @8", O, 256, 8, 512, 4, 2, 1),
    ("CE.c5 512>1024 k4s2 @4", O, 512, 4, 1024, 4, 2, 1),
    ("DEC.c0 192>256 k3 @8", N, 192, 8, 256, 3, 1, 1),
    ("SPADE3.gb 128>128 k3 @64", N, 128, 64, 128, 3, 1, 1),
    ("SPADE2.gb 128>256 k3 @32", N, 128, 32, 256, 3, 1, 1),
    ("SPADE1.gb 128>512 k3 @16", N, 128, 16, 512, 3, 1, 1),
    ("SPADE3.sh 64>128 k3 @64", N, 64, 64, 128, 3, 1, 1),
    ("DEC.dc3(as conv) 64>128 k4s2 @64", N, 64, 64, 128, 4, 2, 1),
    ("DEC.c4 64>3 k7 @64", N, 64, 64, 3, 7, 1, 3),
    ("Dobj.1 box 64>128 k3s2 @33", O, 64, 33, 128, 3, 2, 0),
    ("Dobj.2 box 128>256 k3s2 @17", O, 128, 17, 256, 3, 2, 0),
    ("Dobj.3 box 256>512 k3s2 @9", O, 256, 9, 512, 3, 2, 0),
    ("Dobj.4 box 512>1024 k3s2 @5", O, 512, 5, 1024, 3, 2, 0),
    ("Dobj.4 box 512>1024 k3s2 @5 to2x2", O, 512, 5, 1024, 3, 2, 0),
    ("Dimg.0b 64>64 k3 @64", N, 64, 64, 64, 3, 1, 1),
    ("Dimg.1b 64>128 k3 @32", N, 64, 32, 128, 3, 1, 1),
    ("Dobj.0a 3>64 k3 @32", O, 3, 32, 64, 3, 1, 1),
    ("Dobj.0b 64>64 k3 @32", O, 64, 32, 64, 3, 1, 1),
    ("Dobj.1b 64>128 k3 @32", O, 64, 32, 128, 3, 1, 1),
    ("Dobj.2b 128>256 k3 @16", O, 128, 16, 256, 3, 1, 1),
    ("Dobj.3b 256>512 k3 @8", O, 256, 8, 512, 3, 1, 1),
    ("Dobj.4a 512>512 k3 @4", O, 512, 4, 512, 3, 1, 1),
    ("Dobj.4b 512>1024 k3 @4", O, 512, 4, 1024, 3, 1, 1),
    ("Dobj.1sc 64>128 k1 @16", O, 64, 16, 128, 1, 1, 0),
    ("Dobj.2sc 128>256 k1 @8", O, 128, 8, 256, 1, 1, 0),
    ("Dobj.3sc 256>512 k1 @4", O, 256, 4, 512, 1, 1, 0),
    ("Dobj.4sc 512>1024 k1 @2", O, 512, 2, 1024, 1, 1, 0),
    ("Dobj.0sc 3>64 k1 @32", O, 3, 32, 64, 1, 1, 0),
    ("Dimg.0a 3>64 k3 @64 (B=64)", N, 3, 64, 64, 3, 1, 1),
    ("Dimg.0sc 3>64 k1 @64 (B=64)", N, 3, 64, 64, 1, 1, 0),
    ("128px SPADE5.gb 128>256 k3 @128", 32, 128, 128, 256, 3, 1, 1),
    ("128px DEC.c6 128>128 k5 @128", 32, 128, 128, 128, 5, 1, 2),
    ("128px Dobj.0b 64>64 k3 @64 (B=210)", 210, 64, 64, 64, 3, 1, 1),
    ("128px Dobj.1b 64>128 k3 @64 (B=210)", 210, 64, 64, 128, 3, 1, 1),
]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


sel = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'shape':36s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>6s} | {'bwdD ms':>8s} {'TF/s':>6s} | {'bwdW ms':>8s} {'TF/s':>6s}")
for name, B, Cin, H, Cout, ks, s, p in SHAPES:
    if sel and sel not in name:
        continue
    x = torch.randn(B, Cin, H, H, device=dev)
    w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
    y = L.conv2d_fwd(x, w, None, s, p)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Cin * ks * ks
    tf = timeit(lambda: L.conv2d_fwd(x, w, None, s, p))
    td = timeit(lambda: L.conv2d_bwd_data(dy, w, (H, H), s, p))
    tw = timeit(lambda: L.conv2d_bwd_weight(dy, x, ks, s, p))
    print(f"{name:36s} {fl/1e9:8.1f} | {tf:8.3f} {fl/tf/1e9:6.1f} | {td:8.3f} {fl/td/1e9:6.1f} | {tw:8.3f} {fl/tw/1e9:6.1f}")
