"""Op-level: folded normalise -> conv against the two-pass form in bf16 and split modes: how many outputs differ, by how much, and what the
pre-rounding difference of the transform itself is."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd")]
import torch
from agl import functional as F, lib as L, nn as A
from agl.generator import ConditionalBatchNorm2d
torch.manual_seed(0)
N, Cin, H, Cout = 48, 64, 32, 128
x = torch.randn(N, Cin, H, H) * 1.7 + 0.6
labels = torch.randint(0, 179, (N,))
norm = ConditionalBatchNorm2d(Cin, 179)
conv = A.Conv2d(Cin, Cout, kernel_size=4, stride=2, padding=1, bias=False)
for mode, flags in (("bf16", L.CONV_BF16), ("split", L.CONV_SPLIT3)):
    res = {}
    for folded in (True, False):
        nd, cd = copy.deepcopy(norm).cuda(), copy.deepcopy(conv).cuda()
        F.NORM_FOLD = folded
        with L.conv_flags(flags | L.CONV_ANY_GRID), torch.no_grad():
            res[folded] = F.norm_conv2d(x.cuda(), nd, labels.cuda(), cd, relu=True, training=True).cpu().double()
    F.NORM_FOLD = True
    a, b = res[True], res[False]
    d = (a - b).abs()
    print(f"{mode}: y folded vs two-pass: differing elements {float((d > 0).double().mean()):.3f}, max {float(d.max() / b.abs().max()):.2e}, rms {float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()):.2e} (relative to y's rms)")
# the transform itself on the CPU: fma(x, scale, shift) with double-formed tables vs ((x - mean) * rstd) * gamma + beta in fp32
xm = x.double().mean((0, 2, 3)); xv = x.double().var((0, 2, 3), unbiased=False)
rstd = 1.0 / torch.sqrt(xv + 1e-5)
emb = norm.embed.weight.detach()[labels].double()
gam, bet = emb[:, :Cin], emb[:, Cin:]
scale = (rstd[None] * gam); shift = bet - xm[None] * scale
t_fold = torch.addcmul(shift.float()[:, :, None, None], x, scale.float()[:, :, None, None])
t_two = ((x - xm.float()[None, :, None, None]) * rstd.float()[None, :, None, None]) * gam.float()[:, :, None, None] + bet.float()[:, :, None, None]
exact = (x.double() - xm[None, :, None, None]) * rstd[None, :, None, None] * gam[:, :, None, None] + bet[:, :, None, None]
for nm, t in (("fold", t_fold), ("two-pass", t_two)):
    e = (t.double() - exact).abs() / exact.abs().clamp_min(1e-3)
    print(f"transform {nm}: relative error vs double rms {float(e.pow(2).mean().sqrt()):.2e} max {float(e.max()):.2e}")
rb = lambda t: t.to(torch.bfloat16)
print("bf16 roundings that differ between the two transforms: %.4f of the elements" % float((rb(torch.relu(t_fold)) != rb(torch.relu(t_two))).double().mean()))
