"""CPU: the oracle (oracle/graph.py) reproduces the reference-generated fixtures bit-tightly, i.e. the pin
made in the build container still holds wherever the tests run; plus host-side logic (batch builder,
sequence plan, state_dict layout, closed-form fill)."""
import os

import numpy as np
import pytest
import torch

import oracle.graph as OG
import oracle.step as OS
from oracle.fill import fill_state

T = torch.from_numpy


@pytest.fixture(scope="module")
def ops(golden_dir):
    return np.load(os.path.join(golden_dir, "ops_small.npz"))


def same(a, b, tol=2e-6):
    a, b = a.detach().double(), T(np.asarray(b)).double()
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_crop_fixture(ops):
    feats, boxes = T(ops["crop_feats"]), T(ops["crop_boxes"])
    for tag in ("sorted", "unsorted"):
        o2i = T(ops[f"crop_{tag}_o2i"])
        for HH, WW in ((8, 8), (5, 7), (32, 32)):
            f = feats.clone().requires_grad_(True)
            y = OG.crop_boxes(f, boxes, o2i, HH, WW)
            y.backward(T(ops[f"crop_{tag}_{HH}x{WW}_gy"]))
            same(y, ops[f"crop_{tag}_{HH}x{WW}_y"])
            same(f.grad, ops[f"crop_{tag}_{HH}x{WW}_dfeats"])


def test_crop_closed_form_matches_grid_sample(ops):
    """Independent restatement of the sampling formula the HIP kernel implements."""
    feats, boxes, o2i = ops["crop_feats"], ops["crop_boxes"], ops["crop_sorted_o2i"]
    HH = WW = 8
    _, C, H, W = feats.shape
    out = np.zeros((boxes.shape[0], C, HH, WW), np.float64)

    def lin(j, steps):
        if steps == 1:
            return 1.0, 0.0
        st = np.float32(1.0) / np.float32(steps - 1)
        if j < steps // 2:
            return float(np.float32(1.0) - st * np.float32(j)), float(st * np.float32(j))
        r = np.float32(steps - 1 - j)
        return float(st * r), float(np.float32(1.0) - st * r)

    for b in range(boxes.shape[0]):
        x0, y0, x1, y1 = (2 * boxes[b] - 1).tolist()
        for i in range(HH):
            ws, we = lin(i, HH)
            iy = ((ws * y0 + we * y1 + 1) * H - 1) / 2
            for j in range(WW):
                ws, we = lin(j, WW)
                ix = ((ws * x0 + we * x1 + 1) * W - 1) / 2
                fx, fy = int(np.floor(ix)), int(np.floor(iy))
                for dy, wy in ((0, 1 - (iy - fy)), (1, iy - fy)):
                    for dx, wx in ((0, 1 - (ix - fx)), (1, ix - fx)):
                        yy, xx = fy + dy, fx + dx
                        if 0 <= yy < H and 0 <= xx < W:
                            out[b, :, i, j] += feats[o2i[b], :, yy, xx] * wy * wx
    assert np.abs(out - ops["crop_sorted_8x8_y"]).max() <= 1e-5


def test_condbn_spade_convlstm_fixtures(ops):
    from models.generator_obj_att import ConditionalBatchNorm2d, LayoutConvLSTM, SPADE   # containers only (CPU)
    cbn = ConditionalBatchNorm2d(6, 5)
    P = OS.as_params({"n." + k: v for k, v in fill_state(cbn.state_dict()).items()})
    x = T(ops["cbn_x"]).requires_grad_(True)
    y = OG.cond_bn(P, "n.", x, T(ops["cbn_labels"]), True)
    y.backward(T(ops["cbn_gy"]))
    same(y, ops["cbn_y"]); same(x.grad, ops["cbn_dx"]); same(P["n.embed.weight"].grad, ops["cbn_dembed"])
    same(P["n.bn.running_mean"], ops["cbn_rm1"]); same(P["n.bn.running_var"], ops["cbn_rv1"])
    for _ in range(2):
        OG.cond_bn(P, "n.", T(ops["cbn_x"]) * 1.5 + 0.3, T(ops["cbn_labels"]), True)
    same(P["n.bn.running_var"], ops["cbn_rv3"])
    assert int(P["n.bn.num_batches_tracked"]) == 3
    for S in (8, 16):
        sp = SPADE(16, 64)
        P = OS.as_params({"s." + k: v for k, v in fill_state(sp.state_dict()).items()})
        x, seg = T(ops[f"spade{S}_x"]).requires_grad_(True), T(ops[f"spade{S}_seg"]).requires_grad_(True)
        y = OG.spade(P, "s.", x, seg, True)
        y.backward(T(ops[f"spade{S}_gy"]))
        same(y, ops[f"spade{S}_y"]); same(x.grad, ops[f"spade{S}_dx"]); same(seg.grad, ops[f"spade{S}_dseg"], 1e-5)
        same(P["s.mlp_gamma.weight"].grad, ops[f"spade{S}_dWgamma"], 1e-5)
    cl = LayoutConvLSTM(8, 12, [8, 4, 4], (5, 5))
    P = OS.as_params({"c." + k: v for k, v in fill_state(cl.state_dict()).items()})
    x = T(ops["clstm_x"]).requires_grad_(True)
    y = OG.conv_lstm_fuse(P, "c.", x, T(ops["clstm_o2i"]), (8, 4, 4))
    y.backward(T(ops["clstm_gy"]))
    same(y, ops["clstm_y"]); same(x.grad, ops["clstm_dx"]); same(P["c.cell_list.0.conv.weight"].grad, ops["clstm_dW0"], 1e-5)


def test_discriminator_fixtures(ops):
    from models.discriminator import (AttributeDiscriminator, AttributeDiscriminator128, ImageDiscriminator,
                                      ObjectDiscriminator, ResidualBlock, add_sn)
    m = add_sn(ResidualBlock(8, 16, downsample=True))
    P = OS.as_params({"main.1." + k: v for k, v in fill_state(m.state_dict()).items()})
    x = T(ops["dres_x"]).requires_grad_(True)
    y = OG.d_res_block(P, "main.1.", x, True)
    y.backward(T(ops["dres_gy"]))
    same(y, ops["dres_y"]); same(x.grad, ops["dres_dx"], 1e-5); same(P["main.1.resi.3.weight_orig"].grad, ops["dres_dW3"], 1e-5)
    for k in range(2, 8):                      # spectral-norm state after k training forwards
        with torch.no_grad():
            OG.d_res_block(P, "main.1.", T(ops["dres_x"]), True)
        if k in (3, 7):
            same(P["main.1.resi.3.weight_u"], ops[f"dres_u3_after{k}"]); same(P["main.1.resi.3.weight_v"], ops[f"dres_v3_after{k}"])
    for tag, mod, fn in (("dimg", ImageDiscriminator(conv_dim=8), lambda P, x: OG.image_discriminator(P, x, True)),
                         ("dobj", ObjectDiscriminator(conv_dim=8, n_class=10), lambda P, x: OG.object_discriminator(P, x, True)[1]),
                         ("datt", AttributeDiscriminator(conv_dim=8, n_attribute=12), lambda P, x: OG.attribute_discriminator(P, x, True, False)),
                         ("datt128", AttributeDiscriminator128(conv_dim=8, n_attribute=12), lambda P, x: OG.attribute_discriminator(P, x, True, True))):
        m = add_sn(mod)
        P = OS.as_params(fill_state(m.state_dict()))
        x = T(ops[tag + "_x"]).requires_grad_(True)
        y = fn(P, x)
        y.backward(T(ops[tag + "_gy"]))
        same(y, ops[tag + "_y"], 1e-5); same(x.grad, ops[tag + "_dx"], 1e-5)


def test_state_dict_layout_matches_reference(golden_dir):
    """Keys and order of state_dict() (hence of .parameters(), Adam state and the DP gradient arena) equal the
    reference modules' (names captured from the imported reference into the step fixtures)."""
    from models.generator_obj_att import Generator
    from models.generator_obj_att128 import Generator as Generator128
    from models.discriminator import (AttributeDiscriminator, AttributeDiscriminator128, ImageDiscriminator,
                                      ObjectDiscriminator, add_sn)
    for tag, G, A in (("64", Generator, AttributeDiscriminator), ("128", Generator128, AttributeDiscriminator128)):
        g = np.load(os.path.join(golden_dir, f"step{tag}.npz"))
        nets = {"G": G(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=32, attribute_dim=106),
                "D_img": add_sn(ImageDiscriminator(conv_dim=64)), "D_obj": add_sn(ObjectDiscriminator(n_class=179)),
                "D_att": add_sn(A(n_attribute=106))}
        for k, net in nets.items():
            assert list(net.state_dict().keys()) == [str(s) for s in g[f"s0_statenames_{k}"]], (tag, k)
            assert [n for n, _ in net.named_parameters()] == [str(s) for s in g[f"s0_gradnames_{k}"]], (tag, k)
    assert sum(p.numel() for p in nets["G"].parameters()) == 31480774
    assert sum(p.numel() for p in nets["D_att"].parameters()) == 30175082


def test_oracle_step_reproduces_fixture_losses(golden_dir):
    """One full oracle iteration at 64 px / batch 4 (the reference's CPU-runnable configuration) from the
    closed-form weights must give the reference's 15 logged losses."""
    from models.generator_obj_att import Generator
    from models.discriminator import AttributeDiscriminator, ImageDiscriminator, ObjectDiscriminator, add_sn
    g = np.load(os.path.join(golden_dir, "step64.npz"))
    nets = [Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=32, attribute_dim=106),
            add_sn(ImageDiscriminator(conv_dim=64)), add_sn(ObjectDiscriminator(n_class=179)), add_sn(AttributeDiscriminator(n_attribute=106))]
    ob = OS.OracleBackend(*[fill_state(m.state_dict()) for m in nets], res128=False, obj_size=32)
    b = {k[len("batch_"):]: T(g[k]) for k in g.files if k.startswith("batch_")}
    losses, outs = OS.run_step(ob, b, T(g["pos_weight"]), [T(e) for e in g["s0_eps_d"]], [T(e) for e in g["s0_eps_g"]])
    for name, ref in zip(g["s0_loss_names"], g["s0_loss_values"]):
        assert abs(losses[str(name)] - ref) <= 1e-5 * max(1.0, abs(ref)), (str(name), losses[str(name)], ref)
    same(outs[4], g["s0_out_img_rec"], 1e-5)


def _eval_states(g, res128):
    """Closed-form weights + the advanced spectral-norm u/v the fixture recorded (oracle/make_golden.py::eval_mode)."""
    if res128:
        from models.generator_obj_att128 import Generator
        from models.discriminator import AttributeDiscriminator128 as AttD
    else:
        from models.generator_obj_att import Generator
        from models.discriminator import AttributeDiscriminator as AttD
    from models.discriminator import ImageDiscriminator, ObjectDiscriminator, add_sn
    nets = {"G": Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=64 if res128 else 32, attribute_dim=106),
            "D_img": add_sn(ImageDiscriminator(conv_dim=64)), "D_obj": add_sn(ObjectDiscriminator(n_class=179)),
            "D_att": add_sn(AttD(n_attribute=106))}
    states = {}
    for k, m in nets.items():
        st = fill_state(m.state_dict())
        for name in st:
            key = f"sn_{k}_{name}"
            if key in g.files:
                st[name] = T(g[key])
        states[k] = st
    return nets, states


@pytest.mark.parametrize("tag", ["64", "128"])
def test_oracle_eval_mode_reproduces_reference_fixture(tag, golden_dir):
    """The oracle's train=False branches (BatchNorm running statistics, spectral norm without a power iteration) against
    the reference's netG.eval() / netD.eval() outputs (tests/golden/eval{64,128}.npz, test64.py:96-101,132-141)."""
    g = np.load(os.path.join(golden_dir, f"eval{tag}.npz"))
    res128 = tag == "128"
    _, st = _eval_states(g, res128)
    P = {k: OS.as_params(v) for k, v in st.items()}
    b = {k[len("batch_"):]: T(g[k]) for k in g.files if k.startswith("batch_")}
    eps = [T(e) for e in g["eps"]]
    torch.set_num_threads(8)
    with torch.no_grad():
        out = OG.generator(P["G"], b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], b["z"], b["attribute"],
                           b["masks_shift"], b["boxes_shift"], b["attribute_est"], obj_size=64 if res128 else 32, res128=res128,
                           train=False, eps=eps)
        names = ["crops_input", "crops_input_rec", "crops_rand", "crops_shift", "img_rec", "img_rand", "img_shift", "mu", "logvar",
                 "z_rand_rec", "z_rand_shift"]
        for n, a in zip(names, out):
            same(a, g["out_" + n], 1e-5)
        img, crops = T(g["out_img_rand"]), T(g["out_crops_rand"])
        same(OG.image_discriminator(P["D_img"], img, False), g["d_img"], 1e-5)
        s_, c_ = OG.object_discriminator(P["D_obj"], crops, False)
        same(s_, g["d_obj_src"], 1e-5)
        same(c_, g["d_obj_cls"], 1e-5)
        same(OG.attribute_discriminator(P["D_att"], crops, False, res128), g["d_att"], 1e-5)


def test_synthetic_batch_schema_and_shift_rule():
    from agl import synth
    b = synth.make_batch(6, 64, seed=1)
    O = b["objs"].shape[0]
    assert b["imgs"].shape == (6, 3, 64, 64) and b["masks"].shape == (O, 1, 64, 64) and b["boxes"].shape == (O, 4)
    assert b["obj_to_img"].dtype == np.int64 and np.all(np.diff(b["obj_to_img"]) >= 0)
    counts = np.bincount(b["obj_to_img"])
    assert counts.min() >= 3 and counts.max() <= 9
    assert np.all(b["objs"] >= 1) and np.all(b["objs"] < 179)
    for i in range(O):
        x0, y0, x1, y1 = b["boxes"][i]
        m = np.zeros((64, 64), np.float32)
        m[round(float(y0) * 64):round(float(y1) * 64), round(float(x0) * 64):round(float(x1) * 64)] = 1
        assert np.array_equal(m, b["masks"][i, 0])
        w = x1 - x0
        sx0 = b["boxes_shift"][i, 0]
        if w < 0.5 and x0 > 1 - x1:
            assert abs(sx0 - (x0 - 0.8 * x0)) < 1e-6
        elif w < 0.5 and 1 - x1 > x0:
            assert abs(sx0 - (x0 + 0.8 * (1 - x1))) < 1e-6
    assert np.all(b["attribute_est"].sum(1) >= 1)          # every object has an estimated or annotated attribute
    sh = synth.shard(b, 1, 2)
    assert sh["imgs"].shape[0] == 3 and sh["obj_to_img"].min() == 0 and sh["obj_to_img"].max() == 2


def test_sequence_plan_row_maps():
    from agl.convlstm import SequencePlan
    o2i = torch.tensor([0] * 3 + [1] + [2] * 9 + [3] * 5)
    p = SequencePlan(o2i, "cpu")
    assert (p.O, p.N, p.T) == (18, 4, 9) and p.n_t == [4, 3, 3, 2, 2, 1, 1, 1, 1] and p.off[-1] == 18
    tm = p.tm_to_obj.tolist()
    assert sorted(tm) == list(range(18))
    assert tm[:4] == [4, 13, 0, 3]                       # step 0 of the runs ordered by length 9,5,3,1
    assert p.last_rows.tolist() == [p.off[2] + 2, p.off[0] + 3, p.off[8] + 0, p.off[4] + 1]
    assert p.hprev_rows.tolist()[:3] == [0, 1, 2] and len(p.hprev_rows) == 18 - 4
