"""3x3 input gradient with a ReLU mask: fp32 NCHW / bf16 NCHW / channel-blocked bf16 mask (bf16 arithmetic): us per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
from agl import lib as L
L.set_conv_precision("bf16")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for (N, C, H, Co) in ((210, 64, 64, 64), (210, 128, 32, 128), (32, 64, 128, 64)):
    dy = torch.randn(N, Co, H, H, device="cuda"); w = torch.nn.Parameter(torch.randn(Co, C, 3, 3, device="cuda") * 0.05)
    ws = L.WeightSrc(w, lambda: 0)
    m32 = torch.randn(N, C, H, H, device="cuda"); m16 = m32.to(torch.bfloat16); mb = L.to_blocked_dev(m32)
    out = torch.zeros(N, C, H, H, device="cuda")
    for acc in (False, True):
        r = [t(lambda m=m: L.conv2d_bwd_data(dy, w, (H, H), 1, 1, pos_mask=m, out=out, accumulate=acc, wsrc=ws)) for m in (None, m32, m16, mb)]
        print((N, C, H, Co), "accumulate" if acc else "fresh", "no mask %.1f  fp32 %.1f  bf16 %.1f  blocked %.1f us" % tuple(r))
