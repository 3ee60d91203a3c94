"""CPU coverage of the rows next to the hot path: checkpoint file conventions
(utils/model_saver_iter.py:6-87), the rows the attribute swap touches (train64.py:170-178) and the oracle's
host-logic restatement against hand-computed cases."""
import os
import random

import numpy as np
import torch
import torch.nn as nn


def test_checkpoint_naming_pruning_and_latest(tmp_path):
    from agl import checkpoint as CK
    d = str(tmp_path / "ck")
    m = nn.Sequential(nn.Linear(3, 2), nn.BatchNorm1d(2))
    assert CK.load_model(m, d, appendix="netG", iter='l') == 0          # directory missing -> scratch
    for it in (1000, 2000, 3000, 4000):
        with torch.no_grad():
            m[0].weight.fill_(float(it))
        CK.save_model(m, d, appendix="netG", iter=it, save_num=2, save_step=1000)
        CK.save_model(m, d, appendix="netD_image", iter=it, save_num=2, save_step=1000)
    names = sorted(os.listdir(d))
    assert names == ["iter-3000_netD_image.pkl", "iter-3000_netG.pkl", "iter-4000_netD_image.pkl", "iter-4000_netG.pkl"]
    m2 = nn.Sequential(nn.Linear(3, 2), nn.BatchNorm1d(2))
    assert CK.load_model(m2, d, appendix="netG", iter='l') == 4000
    assert float(m2[0].weight[0, 0]) == 4000.0
    assert CK.load_model(m2, d, appendix="netG", iter=3000) == 3000
    assert float(m2[0].weight[0, 0]) == 3000.0
    assert CK.load_model(m2, d, appendix="netG", iter=1000) == 0        # pruned -> scratch
    assert CK.load_model(m2, d, appendix="netG", iter='s') == 0
    # the file is a plain state_dict (what the reference's torch.save(model.state_dict()) writes)
    sd = torch.load(os.path.join(d, "iter-4000_netG.pkl"))
    assert list(sd.keys()) == list(m.state_dict().keys())


def test_checkpoint_replays_reference_saver_trace(tmp_path, golden_dir):
    """N4 pinned to the reference: tests/golden/saver_trace.json records what the REFERENCE's utils/model_saver_iter.py
    did for a save / prune / load sequence (oracle/make_golden.py::saver_trace drives it); agl.checkpoint must leave the
    same directory listing after every save, read the same file and return the same iteration for every load."""
    import json
    from agl import checkpoint as CK
    ops = json.load(open(os.path.join(golden_dir, "saver_trace.json")))
    d = str(tmp_path / "models")
    picked = []
    real_load = torch.load
    try:
        torch.load = lambda path, *a, **k: (picked.append(os.path.basename(path)), real_load(path, *a, **k))[1]
        for o in ops:
            if o["op"] == "save":
                CK.save_model(nn.Linear(3, 2), d, appendix=o["appendix"], iter=o["iter"], save_num=o["save_num"], save_step=o["save_step"])
                assert sorted(os.listdir(d)) == o["files_after"], o
            else:
                picked.clear()
                r = CK.load_model(nn.Linear(3, 2), d, appendix=o["appendix"], iter=o["iter"])
                assert r == o["returned"], o
                assert (picked[-1] if picked else None) == o["picked"], (o, picked)
    finally:
        torch.load = real_load


def test_latest_model_never_picks_the_optimizer_file(tmp_path):
    """`iter-<n>_optim.pkl` lives next to the network files; load_model(iter='l', appendix=None) and an
    explicit iteration must skip it."""
    from agl import checkpoint as CK
    d = str(tmp_path / "ck")
    m = nn.Linear(3, 2)
    CK.save_model(m, d, appendix=None, iter=1000, save_num=5, save_step=1000)
    torch.save({"g_m": torch.zeros(3), "g_step": 7}, os.path.join(d, "iter-2000_optim.pkl"))
    torch.save({"g_m": torch.zeros(3), "g_step": 7}, os.path.join(d, "iter-1000_optim.pkl"))
    assert CK.load_model(nn.Linear(3, 2), d, appendix=None, iter='l') == 1000
    assert CK.load_model(nn.Linear(3, 2), d, appendix=None, iter=1000) == 1000
    assert CK.load_model(nn.Linear(3, 2), d, appendix=None, iter=2000) == 0      # only an optimiser file: scratch


def test_dropin_import_paths():
    import utils.model_saver_iter as MS
    from agl import checkpoint as CK
    assert MS.load_model is CK.load_model and MS.save_model is CK.save_model
    import models.discriminator as MD
    import models.spade.networks.loss as ML
    from agl import losses as LS
    assert MD.loss_hinge_dis is LS.loss_hinge_dis and ML.loss_hinge_gen is LS.loss_hinge_gen


def test_swap_rows_match_oracle_loop():
    from agl import hostlogic as H, synth
    import oracle.hostlogic as OH
    bn = synth.make_batch(9, 64, seed=3)
    o2i = torch.from_numpy(bn["obj_to_img"])
    A = bn["attribute_gt"].shape[1]
    att = torch.from_numpy(bn["attribute_gt"]).clone()
    est = att.clone()
    matrix = torch.from_numpy(synth.make_cooccurrence())
    before = att.clone()
    random.seed(4)
    OH.swap_attributes(att, est, torch.from_numpy(bn["objs"]), o2i, matrix, 9)
    changed = sorted(int(i) for i in torch.nonzero((att != before).any(dim=1)).view(-1))
    rows = H.swap_rows(o2i, 9)
    assert set(changed) <= set(rows)
    P = np.bincount(bn["obj_to_img"])
    assert len(rows) == sum(int(P[i]) // 2 for i in range(3))
    for r in rows:                                           # new rows: 1 or 2 attributes, none of the old ones
        assert 1 <= int(att[r].sum()) <= 2
        assert float((att[r] * before[r]).sum()) == 0.0
        assert torch.equal(att[r], est[r])
    assert att.shape[1] == A


def test_oracle_estimate_and_deprocess_small_cases():
    import oracle.hostlogic as OH
    attr = torch.zeros(3, 5)
    attr[1, 2] = 1
    logits = torch.tensor([[0.1, 0.9, 0.0, 0.0, 0.0], [5.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 2.0]])
    est = OH.estimate_attributes(logits, attr)
    assert est.tolist() == [[0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 1]]
    x = torch.zeros(1, 3, 2, 2)
    x[0, 0, 0, 0] = 1.0
    b = OH.imagenet_deprocess_batch(x, rescale=True)
    assert b.dtype == torch.uint8 and int(b.max()) == 255 and int(b.min()) == 0
    # no rescale: (x*std + mean)*255 truncated
    b2 = OH.imagenet_deprocess_batch(torch.zeros(1, 3, 1, 1), rescale=False)
    assert b2.view(-1).tolist() == [int(np.float32(0.485) * np.float32(255)), int(np.float32(0.456) * np.float32(255)),
                                    int(np.float32(0.406) * np.float32(255))]


def test_spade_block_class_index_maps():
    """Host-side index maps of the SPADE restructure (agl.functional._grid_map): composing 'up3' -> conv -> '3to5' ->
    conv -> '5tof' must reproduce, row by row, the classes a 3x3 convolution sees on an f-fold nearest up-sampling:
    checked by brute force on 1-D signals with a 3-tap box convolution (the maps act per axis)."""
    from agl import functional as F
    cpu = torch.device("cpu")
    for blocks, f in ((8, 8), (8, 16), (4, 8)):
        g = torch.Generator().manual_seed(f + blocks)
        seg = torch.randn(blocks, generator=g)
        conv = lambda v: torch.nn.functional.conv1d(v.view(1, 1, -1), torch.tensor([[[0.3, -1.1, 0.7]]]), padding=1).view(-1)
        full = seg.repeat_interleave(f)
        a_full = torch.relu(conv(full))
        gb_full = conv(a_full)
        m_up3, lo_up3, _ = F._grid_map("up3", blocks, 0, cpu)
        m35, lo35, _ = F._grid_map("3to5", blocks, 0, cpu)
        m5f, lo5f, src5 = F._grid_map("5tof", blocks, f, cpu)
        m3f, _, _ = F._grid_map("3tof", blocks, f, cpu)
        a3 = torch.relu(conv(seg[m_up3.long()]))
        assert torch.allclose(a3[m3f.long()], a_full, atol=1e-6)
        gb5 = conv(a3[m35.long()])
        assert src5 == 5 * blocks and gb5.numel() == 5 * blocks
        assert torch.allclose(gb5[m5f.long()], gb_full, atol=1e-6)
        for m, lo in ((m_up3, lo_up3), (m35, lo35), (m5f, lo5f)):      # range starts of the inverse map
            mm = m.tolist()
            assert mm == sorted(mm) and lo[0] == 0 and lo[-1] == len(mm)
            for i in range(len(lo) - 1):
                assert all(mm[j] == i for j in range(int(lo[i]), int(lo[i + 1])))


def test_oracle_layout_from_boxes_agrees_with_batch_builder_and_hand_cases():
    """oracle/hostlogic.py::layout_from_boxes (data/vg_custom_mask.py:136-158 restated) against hand-computed cases and against the
    independent restatement inside the synthetic batch builder (agl/synth.py)."""
    import oracle.hostlogic as OH
    from agl import synth
    boxes = torch.tensor([[0.0, 0.0, 1.0, 1.0], [0.25, 0.1, 0.75, 0.9], [0.5, 0.25, 0.75, 0.5], [0.1, 0.0, 0.3, 0.5]])
    bs, m, ms = OH.layout_from_boxes(boxes, 8)
    assert torch.equal(bs[0], boxes[0]) and torch.equal(bs[1], boxes[1])            # wide / equal border distances: no shift
    assert torch.allclose(bs[2], torch.tensor([0.1, 0.25, 0.35, 0.5]))              # left 0.5 > right 0.25: shift left by 0.4
    assert torch.allclose(bs[3], torch.tensor([0.66, 0.0, 0.86, 0.5]))              # right 0.7 > left 0.1: shift right by 0.56
    assert m[2, 0].sum() == 2 * 2 and m[2, 0, 2:4, 4:6].all()                       # rows round(2)..round(4), cols 4..6
    assert ms[2, 0, 2:4, 1:3].all() and ms[2, 0].sum() == 4                         # cols round(0.8)=1 .. round(2.8)=3
    b = synth.make_batch(5, 64, seed=4)
    bs, m, ms = OH.layout_from_boxes(torch.from_numpy(b["boxes"]), 64)
    assert np.array_equal(bs.numpy(), b["boxes_shift"]) and np.array_equal(m.numpy(), b["masks"]) and np.array_equal(ms.numpy(), b["masks_shift"])


def test_oracle_attribute_edit_logic():
    import oracle.hostlogic as OH
    a = torch.zeros(3, 106); a[0, 2] = 1; a[1, 50] = 1; a[2, 95] = 1
    e = a.clone(); e[1, 8] = 1
    a2, e2 = OH.edit_attribute_rows(a, e, 95)
    assert a2[:, 95].all() and a2[0, 2] == 0 and a2[1, 50] == 1 and e2[1, 8] == 0 and e2[1, 50] == 1
    lg = torch.zeros(2, 106); lg[0, :5] = torch.tensor([5., 4., 3., 2., 1.]); lg[1, 95] = 9.
    lg2 = torch.zeros(2, 106); lg2[0, 95] = 7.
    changed, success = OH.edit_success(lg, lg2, 95)
    assert changed == [0] and success == [0]
