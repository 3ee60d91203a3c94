"""Error of the split (AGL_CONV_SPLIT3) convolution kernels against fp64, beside the exact fp32 MFMA kernels: rms / max / mean signed
error per shape, forward, input gradient and weight gradient.  python tools/split_acc.py [scale]  (scale multiplies x: range check)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd")]
import torch
import torch.nn.functional as TF
from agl import lib as L

xs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ws = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
print("split products per MAC:", L.load().agl_conv2d_split_products(), "x scale", xs, "w scale", ws)
CASES = [(9, 512, 4, 768, 3, 1, 1), (9, 512, 8, 768, 3, 1, 1), (2, 512, 16, 768, 3, 1, 1), (4, 64, 32, 128, 3, 1, 1), (64, 64, 32, 64, 3, 1, 1),
         (2, 32, 16, 128, 5, 1, 2), (32, 128, 8, 512, 5, 1, 2), (16, 64, 32, 128, 4, 2, 1), (8, 256, 16, 256, 4, 2, 1), (8, 128, 32, 128, 1, 1, 0),
         (40, 256, 16, 256, 3, 1, 1), (40, 512, 8, 512, 3, 1, 1), (24, 256, 16, 512, 4, 2, 1), (40, 512, 8, 1024, 4, 2, 1), (64, 1024, 4, 1024, 3, 1, 1),
         (64, 128, 32, 256, 4, 2, 1), (48, 512, 8, 512, 5, 1, 2)]
g = torch.Generator().manual_seed(5)
for (N, Cin, H, Cout, ks, st, pad) in CASES:
    x = torch.randn(N, Cin, H, H, generator=g) * xs
    w = torch.randn(Cout, Cin, ks, ks, generator=g) * (ws / (Cin * ks * ks) ** 0.5)
    OH = (H + 2 * pad - ks) // st + 1
    dy = torch.randn(N, Cout, OH, OH, generator=g)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = TF.conv2d(x64, w64, None, stride=st, padding=pad)
    y64.backward(dy.double())
    refs = (y64.detach(), x64.grad, w64.grad)
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    out = {}
    for name, flags in (("exact", 0), ("split", L.CONV_SPLIT3 | L.CONV_ANY_GRID)):
        with L.conv_flags(flags):
            out[name] = (L.conv2d_fwd(xd, wd, None, st, pad), L.conv2d_bwd_data(dyd, wd, (H, H), st, pad), L.conv2d_bwd_weight(dyd, xd, ks, st, pad))
            pipe = L.load().agl_conv2d_last_pipe()
    for i, what in enumerate(("fwd", "dx ", "dw ")):
        ref = refs[i]
        sc = float(ref.abs().max())
        row = []
        for name in ("exact", "split"):
            e = out[name][i].cpu().double() - ref
            row.append("%s rms %.2e max %.2e bias %+.1e" % (name, float(e.pow(2).mean().sqrt()) / sc, float(e.abs().max()) / sc, float(e.mean()) / sc))
        ee, es = (float((out[n][i].cpu().double() - ref).abs().max()) for n in ("exact", "split"))
        print("%-28s %s | %s | %s | ratio %.2f %s" % ((N, Cin, H, Cout, ks, st), what, row[0], row[1], es / max(ee, 1e-30), "" if es <= 2 * ee + 2e-7 * sc else "  <-- over the gate"))
