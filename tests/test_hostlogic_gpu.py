"""GPU parity of the rows next to the hot path against oracle/hostlogic.py."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,rescale", [((5, 3, 64, 64), True), ((3, 3, 128, 128), True), ((4, 3, 32, 32), False),
                                           ((2, 3, 7, 5), True)])
def test_imagenet_deprocess_bytes_equal(shape, rescale):
    from agl import hostlogic as H
    import oracle.hostlogic as OH
    g = torch.Generator().manual_seed(shape[0] * 7 + shape[2])
    x = torch.randn(shape, generator=g) * 1.7
    want = OH.imagenet_deprocess_batch(x, rescale=rescale)
    got = H.imagenet_deprocess_batch(x.to(DEV), rescale=rescale)
    assert got.dtype == torch.uint8 and got.device.type == "cpu"
    assert torch.equal(got, want), int((got.int() - want.int()).abs().max())


def test_dropin_data_utils_path():
    import data.utils as DU
    from agl import hostlogic as H
    assert DU.imagenet_deprocess_batch is H.imagenet_deprocess_batch
    assert DU.INV_IMAGENET_MEAN == [-m for m in DU.IMAGENET_MEAN]


@pytest.mark.parametrize("n_images,seed", [(9, 1), (64, 2), (2, 3)])
def test_attribute_swap_same_draws_as_loop(n_images, seed):
    """Seeded python `random`: the device path must pick the same attributes as the per-object loop."""
    from agl import hostlogic as H, synth
    import oracle.hostlogic as OH
    bn = synth.make_batch(n_images, 64, seed=seed)
    objs = torch.from_numpy(bn["objs"])
    o2i = torch.from_numpy(bn["obj_to_img"])
    matrix = torch.from_numpy(synth.make_cooccurrence())
    att0 = torch.from_numpy(bn["attribute_gt"])
    est0 = torch.from_numpy(bn["attribute_est"])
    a_ref, e_ref = att0.clone(), est0.clone()
    random.seed(100 + seed)
    OH.swap_attributes(a_ref, e_ref, objs, o2i, matrix, n_images)
    tail_ref = random.random()
    a_dev, e_dev = att0.clone().to(DEV), est0.clone().to(DEV)
    random.seed(100 + seed)
    rows = H.swap_attributes(a_dev, e_dev, objs.to(DEV), o2i, matrix, n_images)
    assert random.random() == tail_ref                       # consumed exactly the same number of draws
    assert torch.equal(a_dev.cpu(), a_ref) and torch.equal(e_dev.cpu(), e_ref)
    assert rows.numel() == len(H.swap_rows(o2i, n_images))


def test_estimate_attributes_equals_loop():
    from agl import hostlogic as H
    import oracle.hostlogic as OH
    g = torch.Generator().manual_seed(3)
    O, A = 393, 106
    logits = torch.randn(O, A, generator=g)
    attr = (torch.rand(O, A, generator=g) < 0.01).float()
    want = OH.estimate_attributes(logits, attr)
    got = H.estimate_attributes(logits.to(DEV), attr.to(DEV)).cpu()
    assert torch.equal(got, want)


def test_checkpoint_roundtrip_resumes_training(tmp_path):
    """Save all four networks + the Adam arenas after one iteration, restore into freshly built networks, and the
    second iteration must match the uninterrupted run (weights are views into flat arenas: load_state_dict copies
    in place)."""
    from agl import checkpoint as CK, synth
    from agl.trainer import Trainer, batch_to_device
    from test_model_gpu import build_nets
    d = str(tmp_path / "models")
    pw = torch.from_numpy(synth.make_pos_weight())
    bn = synth.make_batch(3, 64, seed=21, objs_per_image=[3, 2, 4])
    b = batch_to_device(bn, DEV)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(9)
    eps = [[torch.randn(O, 64, generator=gen) for _ in range(3)] for _ in range(4)]
    names = ("netG", "netD_image", "netD_object", "netD_att")

    nets = build_nets(False)
    tr = Trainer(*nets, pw)
    tr.step(b, eps[0], eps[1])
    for net, nm in zip(nets, names):
        CK.save_model(net, d, appendix=nm, iter=1, save_num=5, save_step=1)
    CK.save_optimizer(tr, d, 1)
    tr.step(b, eps[2], eps[3])
    tr.finish()
    want_loss = tr.loss_dict()
    want = [{k: v.detach().cpu().clone() for k, v in n.state_dict().items()} for n in nets]

    nets2 = build_nets(False)
    for n in nets2:                                            # scramble so that loading matters
        for p in n.parameters():
            p.data.mul_(0.5)
    tr2 = Trainer(*nets2, pw)
    for net, nm in zip(nets2, names):
        assert CK.load_model(net, d, appendix=nm, iter='l') == 1
    assert CK.load_optimizer(tr2, d, 1)
    assert tr2.flat_g.step_count == 1 and tr2.flat_d.step_count == 1
    p0 = next(nets2[0].parameters())
    assert p0.data_ptr() == tr2.flat_g.p.data_ptr()            # still a view of the arena
    tr2.step(b, eps[2], eps[3])
    tr2.finish()
    got_loss = tr2.loss_dict()
    for k, v in want_loss.items():
        assert abs(got_loss[k] - v) <= 1e-5 * max(1.0, abs(v)), (k, got_loss[k], v)
    for n, w in zip(nets2, want):
        for k, v in n.state_dict().items():
            a, r = v.detach().cpu().double(), w[k].double()
            if v.is_floating_point():
                # crop backward accumulates with float atomics (order varies run to run); Adam can flip the sign of
                # noise-level gradients, so compare at 2*lr granularity
                assert float((a - r).abs().max()) <= 4.1e-4 + 1e-5 * float(r.abs().max()), k
            else:
                assert torch.equal(v.cpu(), w[k]), k


def test_layout_tensors_from_boxes_on_device():
    """masks, shifted boxes and shifted masks built in HBM from the boxes alone (agl_layout_from_boxes) equal the
    restatement of data/vg_custom_mask.py:136-158 in oracle/hostlogic.py (python floats, python round, slice clipping) bit for bit,
    including boxes wider than half the image (no shift), equal border distances (no shift), boxes touching the borders and
    shifted boxes that leave the image."""
    import oracle.hostlogic as OH
    from agl import lib as L
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(200, generator=g) * 0.7
    y0 = torch.rand(200, generator=g) * 0.7
    boxes = torch.stack([x0, y0, (x0 + 0.05 + torch.rand(200, generator=g) * 0.6).clamp(max=1.0),
                         (y0 + 0.05 + torch.rand(200, generator=g) * 0.5).clamp(max=1.0)], 1)
    boxes[:6] = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.25, 0.1, 0.75, 0.9], [0.3, 0.2, 0.7, 0.4], [0.0, 0.5, 0.2, 1.0],
                              [0.8, 0.0, 1.0, 0.3], [0.05, 0.05, 0.15, 0.15]])
    for R in (64, 128):
        bs_ref, m_ref, ms_ref = OH.layout_from_boxes(boxes, R)
        bs, m, ms = L.layout_from_boxes(boxes.to(DEV), R)
        assert torch.equal(bs.cpu(), bs_ref)
        assert torch.equal(m.cpu(), m_ref)
        assert torch.equal(ms.cpu(), ms_ref)


def test_attribute_editing_loop_vs_oracle():
    """one iteration of the attribute-editing inference loop (test64.py:114-198) on the eval-mode kernels against
    the oracle graph + the loop's attribute logic restated in oracle/hostlogic.py: attribute estimate and edited attribute
    rows exactly, generated images within 1e-3 relative-to-max, classifier predictions / top-k success lists equal wherever the
    oracle's decision margin exceeds the numerical tolerance."""
    import oracle.graph as OG, oracle.step as OS, oracle.hostlogic as OH
    from oracle.fill import fill_state
    from agl import synth
    from agl.infer import edit_attributes_batch
    from models.generator_obj_att import Generator
    from models.discriminator import AttributeDiscriminator, add_sn
    G = Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=32, attribute_dim=106)
    Da = add_sn(AttributeDiscriminator(n_attribute=106))
    for m in (G, Da):
        m.load_state_dict(fill_state(m.state_dict()))
    Pg = OS.as_params({k: v.clone() for k, v in G.state_dict().items()})
    Pa = OS.as_params({k: v.clone() for k, v in Da.state_dict().items()})
    G.to(DEV), Da.to(DEV)
    bn = synth.make_batch(3, 64, seed=23, objs_per_image=[4, 3, 5])
    b = {k: torch.from_numpy(v) for k, v in bn.items()}
    O = b["objs"].shape[0]
    gen = torch.Generator().manual_seed(2)
    z, z2 = torch.randn(O, 64, generator=gen), torch.randn(O, 64, generator=gen)
    eps = [torch.randn(O, 64, generator=gen) for _ in range(6)]
    d = {k: (v.to(DEV) if k != "obj_to_img" else v) for k, v in b.items()}
    d["attribute"] = d["attribute_gt"].clone()
    Da.eval()                 # state-free classifier for the comparison below (the oracle calls use train=False)
    G.train()
    res = edit_attributes_batch(G, Da, d, tgt=95, z=z, z_edit=z2, eps=eps[:3], eps_edit=eps[3:])
    torch.cuda.synchronize()
    assert G.training and not Da.training, "module modes must be left as the caller set them"
    # ---- oracle
    with torch.no_grad():
        attr = b["attribute_gt"]
        crops = OG.crop_boxes(b["imgs"], b["boxes"], b["obj_to_img"], 32)
        est = OH.estimate_attributes(OG.attribute_discriminator(Pa, crops, False, False), attr)
        gen_o = lambda a, ae, zz, ee: OG.generator(Pg, b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], zz, a,
                                                  b["masks_shift"], b["boxes_shift"], ae, obj_size=32, res128=False, train=False, eps=ee)
        out = gen_o(attr, est, z, eps[:3])
        lg = OG.attribute_discriminator(Pa, out[2], False, False)
        a2, e2 = OH.edit_attribute_rows(attr, est, 95)
        out2 = gen_o(a2, e2, z2, eps[3:])
        lg2 = OG.attribute_discriminator(Pa, out2[2], False, False)
        changed, success = OH.edit_success(lg, lg2, 95)
    assert torch.equal(res["attribute_est"].cpu(), est)
    assert torch.equal(res["attribute_edit"].cpu(), a2) and torch.equal(res["attribute_est_edit"].cpu(), e2)
    scale = float(out[5].abs().max())
    assert float((res["img_rand"].cpu() - out[5]).abs().max()) <= 1e-3 * scale
    assert float((res["img_rand_edit"].cpu() - out2[5]).abs().max()) <= 1e-3 * float(out2[5].abs().max())
    # decisions: compare where the oracle's margin is clear of the logits' numerical tolerance
    tol = 2e-3 * float(lg.abs().max())
    sig = torch.sigmoid(lg)
    clear = (lg - float(torch.logit(torch.tensor(0.9)))).abs() > tol
    assert torch.equal(res["pred"].cpu().bool()[clear], (sig > 0.9)[clear])
    ch = res["changed"].cpu().bool()
    for i in range(O):
        kth5 = lg[i].topk(6)[0]
        margin5 = min(abs(float(lg[i, 95] - kth5[4])), abs(float(lg[i, 95] - kth5[5])))
        if margin5 > tol:
            assert bool(ch[i]) == (i in changed), i
    img_ref = OH.imagenet_deprocess_batch(out[5])
    assert int((res["images"]["rand"].cpu().int() - img_ref.int()).abs().max()) <= 1


def test_attribute_editing_loop_keeps_reference_module_modes():
    """test64.py:114 puts only netG in eval mode: a train-mode netD_att runs one power iteration in each of its four forward
    calls per batch (:129, :146, :180, :183).  The loop must make exactly those four calls and leave both modules' modes alone."""
    import oracle.graph as OG, oracle.step as OS
    from oracle.fill import fill_state
    from agl import synth
    from agl.infer import edit_attributes_batch
    from models.generator_obj_att import Generator
    from models.discriminator import AttributeDiscriminator, add_sn
    G = Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=32, attribute_dim=106)
    Da = add_sn(AttributeDiscriminator(n_attribute=106))
    for m in (G, Da):
        m.load_state_dict(fill_state(m.state_dict()))
    Pa = OS.as_params({k: v.clone() for k, v in Da.state_dict().items()})
    G.to(DEV), Da.to(DEV)
    bn = synth.make_batch(2, 64, seed=5, objs_per_image=[3, 2])
    b = {k: torch.from_numpy(v) for k, v in bn.items()}
    d = {k: (v.to(DEV) if k != "obj_to_img" else v) for k, v in b.items()}
    d["attribute"] = d["attribute_gt"].clone()
    G.train(), Da.train()
    edit_attributes_batch(G, Da, d, tgt=95)
    torch.cuda.synchronize()
    assert G.training and Da.training
    # the oracle's spectral-norm state after four training forwards of the attribute discriminator (inputs do not matter)
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        for _ in range(4):
            OG.attribute_discriminator(Pa, x, True, False)
    sd = Da.state_dict()
    for k in sd:
        if k.endswith("weight_u") or k.endswith("weight_v"):
            assert torch.allclose(sd[k].cpu(), Pa[k], atol=2e-5), k
