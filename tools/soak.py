"""Soak run: N training iterations on one synthetic batch stream; prints losses and peak device memory (GPU box only)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
from bench import build_nets
from agl import synth
from agl.trainer import Trainer, batch_to_device
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = sys.argv[2] if len(sys.argv) > 2 else "f32x3"          # conv arithmetic: f32 | f32x3 | bf16
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 48        # images per iteration (object counts change every iteration)
res = int(sys.argv[4]) if len(sys.argv) > 4 else 64           # 64 | 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
nets, _ = build_nets(res, dev)
tr = Trainer(*nets, torch.from_numpy(synth.make_pos_weight()), conv_dtype=mode)
mem = []
for i in range(steps):
    b = batch_to_device(synth.make_batch(batch - (i % 5), res, seed=1000 + i), dev)      # a new batch (new image / object counts) every iteration
    tr.step(b)
    tr.finish()
    if i % 5 == 0 or i == steps - 1:
        d = tr.loss_dict()
        assert all(math.isfinite(v) for v in d.values()), (i, d)
        mem.append(torch.cuda.max_memory_allocated() / 2**30)
        print(f"iter {i:3d}  D/loss {d['D/loss']:.4f}  G/loss {d['G/loss']:.4f}  peak mem {mem[-1]:.2f} GiB", flush=True)
assert mem[-1] <= mem[1] * 1.25 + 0.5, f"device memory keeps growing: {mem}"
print("soak ok")
