"""Host-side cost of one training iteration (GPU box): with a tiny batch the kernels are short, so the step time is the
time the Python host needs to enqueue the ~2500 C-ABI calls / ~4600 kernel launches of an iteration."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
import bench
from agl import lib as L, synth
from agl.trainer import Trainer, batch_to_device

res = int(os.environ.get("RES", "64"))
per = int(os.environ.get("BATCH", "2"))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
nets, _ = bench.build_nets(res, dev)
tr = Trainer(*nets, torch.from_numpy(synth.make_pos_weight()), estimate_attributes=True)
bn = synth.make_batch(per, res, seed=1)
b = batch_to_device(bn, dev)
O = bn["objs"].shape[0]
eps = [torch.randn(O, 64).to(dev) for _ in range(6)]
for _ in range(3):
    tr.step(b, eps[:3], eps[3:])
torch.cuda.synchronize()
n = 10
c0 = L.CALL_COUNT
t0 = time.perf_counter()
for _ in range(n):
    tr.step(b, eps[:3], eps[3:])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"res {res} batch {per} (O={O}): host enqueue {1e3*(t1-t0)/n:.1f} ms/step, wall {1e3*(t2-t0)/n:.1f} ms/step, "
      f"{(L.CALL_COUNT-c0)/n:.0f} ABI calls/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(b, eps[:3], eps[3:])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
