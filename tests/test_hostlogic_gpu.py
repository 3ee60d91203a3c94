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
