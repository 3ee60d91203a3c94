"""Probe the conv kernel's core loop on GEMM-like shapes (1x1 convs) to separate gather cost from loop structure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
from agl import lib as L
dev = "cuda:0"
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (B, Cin, H, Cout, ks, pad) in [(8, 512, 64, 512, 1, 0), (8, 1024, 64, 1024, 1, 0), (32, 256, 64, 256, 1, 0), (8, 512, 64, 512, 3, 1), (8, 2048, 32, 2048, 1, 0)]:
    x = torch.randn(B, Cin, H, H, device=dev); w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05
    y = L.conv2d_fwd(x, w, None, 1, pad); dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * Cin * ks * ks
    tf = timeit(lambda: L.conv2d_fwd(x, w, None, 1, pad))
    td = timeit(lambda: L.conv2d_bwd_data(dy, w, (H, H), 1, pad))
    tw = timeit(lambda: L.conv2d_bwd_weight(dy, x, ks, 1, pad))
    print(f"B{B} {Cin}>{Cout} k{ks} @{H}: {fl/1e9:8.1f} GF | fwd {tf:7.3f} ms {fl/tf/1e9:6.1f} TF | bwdD {td:7.3f} ms {fl/td/1e9:6.1f} TF | bwdW {tw:7.3f} ms {fl/tw/1e9:6.1f} TF")
if os.environ.get('AGL_LIBRARY'): sys.exit(0)
# practical ceiling: the vendor library's fp32 GEMM on the same box (reference point only, never on the product path)
torch.backends.cuda.matmul.allow_tf32 = False
for (m, n, k) in [(512, 32768, 512), (1024, 32768, 1024), (4096, 4096, 4096), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev)
    t = timeit(lambda: torch.matmul(a, b))
    print(f"torch.matmul fp32 {m}x{n}x{k}: {t:7.3f} ms {2.0*m*n*k/t/1e9:6.1f} TF")
