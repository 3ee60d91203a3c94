"""HIP against HIP: generator outputs / losses of one 128 px bf16 iteration with the bf16-only code paths switched on and off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from agl import functional as F, synth, dtrunk
from agl.trainer import Trainer, batch_to_device
import test_model_gpu as T
pw = torch.from_numpy(synth.make_pos_weight())
bn = synth.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 8, 128, seed=77)
O = bn["objs"].shape[0]
gen = torch.Generator().manual_seed(5)
eps = [torch.randn(O, 64, generator=gen) for _ in range(6)]
def run(fold, y16, trunk=True, dtype="bf16"):
    F.NORM_FOLD, F.SPADE_Y16, dtrunk.D_TRUNK = fold, y16, trunk
    nets = T.build_nets(True)
    tr = Trainer(*nets, pw, conv_dtype=dtype)
    tr.step(batch_to_device(bn, "cuda:0"), eps[:3], eps[3:]); tr.finish(); torch.cuda.synchronize()
    return tr.loss_dict(), [t.detach().cpu().double() for t in tr.last_outputs]
base = run(True, True)
for name, cfg in (("same again", (True, True)), ("fold off", (False, True)), ("y16 off", (True, False)), ("both off", (False, False)), ("f32x3", (True, True, True, "f32x3"))):
    l, o = run(*cfg)
    errs = ["%.1e/%.1e" % (float((a - b).abs().max() / b.abs().max()), float((a - b).pow(2).mean().sqrt() / b.abs().max())) for a, b in zip(o, base[1])]
    print("%-10s max/rms rel-to-max per output: %s" % (name, " ".join(errs)))
    print("           D/loss %.3e  G/loss %.3e" % (abs(l["D/loss"] - base[0]["D/loss"]) / abs(base[0]["D/loss"]), abs(l["G/loss"] - base[0]["G/loss"]) / abs(base[0]["G/loss"])))
