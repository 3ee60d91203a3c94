"""Per-tensor gradient agreement between two arithmetic modes of the SAME iteration on the GPU (no oracle, seconds):
    python tools/grad_profile.py [--res 128] [--batch 8] [--a f32x3] [--b bf16] [--keep-d]
Both runs start from one initial state; the discriminators' Adam update is skipped (unless --keep-d) so that the generator's
gradient — which flows through the discriminators — measures the kernels, not the lr * sign(g) flips of the first Adam step.
Prints, in parameter order, |g_b - g_a| / |g_a| for every tensor of the four networks (D gradients from the D step, G from the G step)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch

import bench
from agl import synth
from agl.trainer import Trainer, batch_to_device

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=128)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--a", default="f32x3")
ap.add_argument("--b", default="bf16")
ap.add_argument("--keep-d", action="store_true")
ap.add_argument("--env-b", default="", help="comma-separated NAME=VALUE switches of agl.functional set for run b only (e.g. NORM_FOLD=0)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
nets, _ = bench.build_nets(a.res, dev)
init = [{k: v.clone() for k, v in m.state_dict().items()} for m in nets]
bn = synth.make_batch(a.batch, a.res, seed=1234)
b = batch_to_device(bn, dev)
O = int(bn["objs"].shape[0])
g = torch.Generator().manual_seed(3)
eps = [torch.randn(O, 64, generator=g).to(dev) for _ in range(6)]
pw = torch.from_numpy(synth.make_pos_weight())
names = ["G", "D_img", "D_obj", "D_att"]


def run(dtype, switches=""):
    from agl import functional as F
    saved = {}
    for kv in filter(None, switches.split(",")):
        k, v = kv.split("=")
        saved[k] = getattr(F, k)
        setattr(F, k, v not in ("0", "False"))
    try:
        for m, st in zip(nets, init):
            m.load_state_dict(st)
        tr = Trainer(*nets, pw, conv_dtype=dtype)
        grads = {}
        if not a.keep_d:
            tr.flat_d.adam_step = lambda *args, **kw: None
        tr.on_d_backward = lambda t: grads.update({n: [q.grad.detach().clone() for q in m.parameters()] for n, m in zip(names[1:], nets[1:])})
        tr.on_g_backward = lambda t: grads.update({"G": [q.grad.detach().clone() for q in nets[0].parameters()]})
        tr.step(b, eps[:3], eps[3:])
        tr.finish()
        torch.cuda.synchronize()
        return grads, tr.loss_dict()
    finally:
        for k, v in saved.items():
            setattr(F, k, v)


ga, la = run(a.a)
gb, lb = run(a.b, a.env_b)
print(f"{a.res} px, batch {a.batch}, O = {O}: {a.b} ({a.env_b}) against {a.a}; D Adam {'kept' if a.keep_d else 'skipped'}")
for k in la:
    print(f"  loss {k:28s} {la[k]:12.6f} {lb[k]:12.6f}  rel {abs(la[k] - lb[k]) / max(1.0, abs(la[k])):.2e}")
for n, m in zip(names, nets):
    big = max(float(t.norm()) for t in ga[n])
    for (pn, _), x, y in zip(m.named_parameters(), ga[n], gb[n]):
        nx = float(x.norm())
        rel = float((y - x).norm()) / (nx + 1e-30)
        print(f"  {n:6s} {pn:52s} |g| {nx:10.3e}  rel L2 {rel:9.2e}{'   (small)' if nx < 1e-2 * big else ''}")
