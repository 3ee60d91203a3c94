"""Stage-by-stage comparison of the HIP generator against the CPU oracle (debug aid, GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd"))
import torch
import oracle.graph as OG, oracle.step as OS
from oracle.fill import fill_state
from agl import synth, functional as F
from agl.convlstm import SequencePlan

res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
res128 = res == 128
if res128:
    from models.generator_obj_att128 import Generator
else:
    from models.generator_obj_att import Generator
DEV = "cuda:0"
G = Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=64 if res128 else 32, attribute_dim=106)
G.load_state_dict(fill_state(G.state_dict())); G.to(DEV)
P = OS.as_params({k: v.cpu() for k, v in G.state_dict().items()})
b = {k: torch.from_numpy(v) for k, v in synth.make_batch(2, res, seed=5, objs_per_image=[2, 3]).items()}
d = {k: (v.to(DEV) if k != "obj_to_img" else v) for k, v in b.items()}
O = b["objs"].shape[0]
eps = torch.randn(O, 64, generator=torch.Generator().manual_seed(3))

def cmp(name, a, r):
    a, r = a.detach().cpu().double(), r.detach().double()
    print(f"{name:34s} rel-to-max {float((a-r).abs().max())/max(float(r.abs().max()),1e-9):.3e}  max|ref| {float(r.abs().max()):.3e}")

with torch.no_grad():
    s = G.obj_size
    co = OG.crop_boxes(b["imgs"], b["boxes"], b["obj_to_img"], s)
    cg = F.crop_boxes(d["imgs"], d["boxes"], d["obj_to_img"].to(DEV), s)
    cmp("crop", cg, co)
    # crop encoder stage by stage
    xo, xg = co, cg
    ce = G.crop_encoder
    for conv, bn, st, pd in (("c1", "bn1", 1, 3), ("c2", "bn2", 2, 1), ("c3", "bn3", 2, 1), ("c4", "bn4", 2, 1), ("conv5", "bn5", 2, 1)):
        xo = torch.nn.functional.conv2d(xo, P["crop_encoder." + conv + ".weight"], None, stride=st, padding=pd)
        xg = getattr(ce, conv)(xg)
        cmp("crop_encoder." + conv, xg, xo)
        xo = torch.relu(OG.cond_bn(P, "crop_encoder." + bn + ".", xo, b["objs"], True))
        xg = getattr(ce, bn)(xg, d["objs"], relu=True)
        cmp("crop_encoder." + bn, xg, xo)
    z_o, mu_o, lv_o = OG.crop_encoder(P, "crop_encoder.", co, b["objs"], True, eps)
    z_g, mu_g, lv_g = ce(cg, d["objs"], eps)
    cmp("mu", mu_g, mu_o); cmp("logvar", lv_g, lv_o); cmp("z", z_g, z_o)
    ao = OG.attribute_encoder(P, "attribute_encoder.", b["objs"], b["attribute"], True)
    ag = G.attribute_encoder(d["objs"], d["attribute"])
    cmp("attribute_encoder", ag, ao)
    # layout encoder pieces (feed identical inputs = oracle values)
    le = G.layout_encoder
    v = torch.cat((ao, z_o), 1)
    ho = v[:, :, None, None] * b["masks"]
    ho = torch.nn.functional.conv2d(ho, P["layout_encoder.c0.weight"], None, padding=1)
    vg = v.to(DEV)
    hg = F.mask_outer(F.linear(vg, le.c0.weight.view(64, -1)), d["masks"], 1)
    cmp("layout c0", hg, ho)
    for conv, bn, relu in ((None, "bn1", True), ("c2", "bn2", True), ("c3", "bn3", True), ("c4", "bn4", False)):
        if conv:
            ho = torch.nn.functional.conv2d(ho, P["layout_encoder." + conv + ".weight"], None, stride=2, padding=1)
            hg = getattr(le, conv)(hg)
            cmp("layout " + conv, hg, ho)
        ho = OG.cond_bn(P, "layout_encoder." + bn + ".", ho, b["objs"], True)
        ho = torch.relu(ho) if relu else ho
        hg = getattr(le, bn)(hg, d["objs"], relu=relu)
        cmp("layout " + bn, hg, ho)
    if res128:
        ho = torch.nn.functional.adaptive_avg_pool2d(ho, 8); hg = F.avg_pool2(hg); cmp("layout pool", hg, ho)
    lo = OG.conv_lstm_fuse(P, "layout_encoder.clstm.", ho, b["obj_to_img"])
    lg = le.clstm(ho.to(DEV), b["obj_to_img"])
    cmp("clstm (same input)", lg, lo)
    ro, rg = lo, lo.to(DEV)
    for r in range(6):
        rp = f"layout_encoder.residual.{r}.main."
        t = torch.nn.functional.conv2d(ro, P[rp + "0.weight"], None, padding=1)
        t = torch.relu(OG._bn(P, rp + "1.", t, True, True))
        t = torch.nn.functional.conv2d(t, P[rp + "3.weight"], None, padding=1)
        ro = ro + OG._bn(P, rp + "4.", t, True, True)
        rg = le.residual[r](rg)
        cmp(f"residual {r}", rg, ro)
    go = OG.global_encoder(P, "global_encoder.", ro, True)
    gg = G.global_encoder(ro.to(DEV))
    cmp("global (same input)", gg, go)
    do = OG.decoder(P, "decoder.", ro, go, True, res128)
    dg = G.decoder(ro.to(DEV), go.to(DEV))
    cmp("decoder (same input)", dg, do)
