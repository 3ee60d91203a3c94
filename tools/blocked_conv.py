"""Channel-blocked prototype against the NCHW forms of the same 3x3 convolution (bf16 arithmetic): tools/blocked_conv.py N Cin H Cout"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
from agl import lib as L
N, Cin, H, Cout = (int(v) for v in sys.argv[1:5])
L.set_conv_precision("bf16")
x = torch.randn(N, Cin, H, H, device="cuda:0"); w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device="cuda:0") * 0.05)
ws = L.WeightSrc(w, lambda: 0)
x16, xb = x.to(torch.bfloat16), L.to_blocked(x)
forms = {"fp32 in / fp32 out (NCHW)": lambda: L.conv2d_fwd(x, w, None, 1, 1, wsrc=ws),
         "bf16 in / bf16 out (NCHW)": lambda: L.conv2d_fwd(x16, w, None, 1, 1, wsrc=ws, out_bf16=True),
         "bf16 in / bf16 out (channel-blocked)": lambda: L.conv2d_fwd_blocked(xb, w, None, wsrc=ws)}
for name, f in forms.items():
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("%s 3x3: %-40s %7.1f us per call (packed weights cached)" % (sys.argv[1:5], name, 1e3 * e0.elapsed_time(e1) / 20))
