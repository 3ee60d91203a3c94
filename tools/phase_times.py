"""Where one iteration spends its GPU time, without a profiler: the trainer records an event at the end of every phase (and of
every discriminator chain, on the chain's stream); this prints when each was reached, in ms after the iteration's start.
    python tools/phase_times.py [--res 128] [--dtype bf16] [--steps 5]      (GPU box only)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch

import bench
from agl import synth
from agl.trainer import Trainer, batch_to_device

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=64)
ap.add_argument("--dtype", default=None)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dtype = a.dtype or ("f32x3" if a.res == 64 else "bf16")
dev = torch.device("cuda:0")
torch.manual_seed(0)
nets, _ = bench.build_nets(a.res, dev)
tr = Trainer(*nets, torch.from_numpy(synth.make_pos_weight()), estimate_attributes=True, conv_dtype=dtype)
per_gpu = 64 if a.res == 64 else 32
bn = synth.make_batch(per_gpu, a.res, seed=1234)
b = batch_to_device(bn, dev)
O = int(bn["objs"].shape[0])
eps = [torch.randn(O, 64).to(dev) for _ in range(6)]
G = tr.netG


def hook(t, label):
    if tr.marks is not None and torch.is_tensor(t) and t.requires_grad:
        t.register_hook(lambda g, label=label: tr._mark("   backward: gradient of " + label + " complete (its producer's stream)"))


# gradient-arrival marks inside the G step's backward (tensor hooks run on the stream of the node that consumes the gradient)
_first, _second, _many = tr._gen_first, tr._gen_second, G.layout_encoder.forward_many


def gen_first(b_, e_):
    out, state = _first(b_, e_)
    sh = state[0]
    for k in ("img_rand", "img_shift", "crops_rand", "crops_shift", "mu_rand", "mu_shift", "mu", "objs_att", "objs_att_est"):
        hook(sh.get(k), k)
    return out, state


def gen_second(state, e_):
    out = _second(state, e_)
    hook(out[4], "img_rec")
    hook(out[1], "crops_rec")
    return out


def forward_many(*args, **kw):
    hs = _many(*args, **kw)
    for i, h in enumerate(hs):
        hook(h, "ConvLSTM output %d (rand / shift)" % i)
    return hs


tr._gen_first, tr._gen_second, G.layout_encoder.forward_many = gen_first, gen_second, forward_many
for _ in range(3):
    tr.step(b, eps[:3], eps[3:])
tr.finish()
torch.cuda.synchronize()
acc = {}
order = []
for it in range(a.steps):
    tr.marks = []
    tr.step(b, eps[:3], eps[3:])
    tr.finish()
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
    m, tr.marks = tr.marks + [("end (Adam of G joined)", end, time.perf_counter())], None
    t0, h0 = m[0][1], m[0][2]
    for label, ev, th in m:
        if label not in acc:
            acc[label] = [0.0, 0.0]
            order.append(label)
        acc[label][0] += t0.elapsed_time(ev) / a.steps
        acc[label][1] += 1e3 * (th - h0) / a.steps
print(f"{a.res} px, {dtype}, batch {per_gpu}, O = {O}: event reached at (ms after the start of the iteration; mean of {a.steps}; the host runs "
      f"ahead, so consecutive iterations overlap at the edges)")
print("       GPU      host (ms after the host entered the iteration: when the mark was ISSUED; the last line includes the final synchronize)")
for label in order:
    print(f"  {acc[label][0]:8.2f}  {acc[label][1]:8.2f}  {label}")
