"""GPU parity of every HIP op (through the C ABI) against the same op evaluated by PyTorch on the CPU
(the arithmetic the reference runs) and against the reference-generated fixtures in
tests/golden/ops_small.npz.  Tolerances are fp32: 2e-5 relative-to-max on forward values, 1e-4 on
gradients (different summation order; atomics in the crop scatter)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def dev(t):
    return t.to(DEV)


def close(a, b, tol, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{what}: rel-to-max err {err:.3e} > {tol:.1e}"


def rn(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g)


CONV_CASES = [
    # N, Cin, H, W, Cout, ks, stride, pad
    (2, 3, 32, 32, 64, 7, 1, 3),      # crop encoder c1 / decoder c5 shape class
    (3, 64, 18, 18, 128, 4, 2, 1),    # k4 s2 even
    (2, 128, 33, 33, 96, 4, 2, 1),    # k4 s2 odd input (LayoutEncoder c3: 33 -> 16)
    (5, 192, 8, 8, 256, 3, 1, 1),     # decoder c0_new
    (4, 72, 8, 8, 48, 5, 1, 2),       # ConvLSTM-like k5
    (2, 64, 16, 16, 3, 7, 1, 3),      # tiny Cout (decoder c4)
    (3, 40, 9, 7, 24, 1, 1, 0),       # 1x1
    (2, 16, 6, 6, 20, 1, 1, 1),       # 1x1 with padding (LayoutEncoder c0)
    (70, 170, 1, 1, 128, 1, 1, 0),    # Linear as 1x1
    (37, 130, 1, 1, 1, 1, 1, 0),      # ... one output (the discriminators' real / fake heads); rows and features off the block sizes
    (9, 64, 1, 1, 2048, 1, 1, 0),     # ... wide (crop encoder fc)
    (1, 3, 10, 12, 8, 3, 1, 1),       # first D conv, ragged
    (3, 64, 17, 17, 128, 3, 2, 0),    # 3x3 stride 2 (box form of conv3x3 + avg-pool): phases of 4:2:2:1 taps
    (40, 32, 5, 5, 64, 3, 2, 0),      # ... on a small map (tap-proportional splits)
    (2, 16, 9, 7, 24, 3, 2, 1),       # ... padded, ragged
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(case):
    from agl import functional as F
    N, Cin, H, W, Cout, ks, s, p = case
    x, w, b = rn(N, Cin, H, W), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = TF.conv2d(xr, wr, br, stride=s, padding=p)
    gy = rn(*yr.shape, seed=3)
    yr.backward(gy)
    xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
    yg = F.conv2d(xg, wg, bg, s, p)
    yg.backward(dev(gy))
    close(yg, yr, 2e-5, "y")
    close(xg.grad, xr.grad, 1e-4, "dx")
    close(wg.grad, wr.grad, 1e-4, "dw")
    close(bg.grad, br.grad, 1e-4, "db")


def test_conv2d_fusions():
    """input ReLU, output ReLU, folded nearest up-sampling, accumulate-into-addend."""
    from agl import functional as F
    x, w, b = rn(3, 12, 8, 8), rn(20, 12, 3, 3, seed=1) * 0.1, rn(20, seed=2)
    add = rn(3, 20, 16, 16, seed=5)
    xr, wr, br, ar = (t.clone().requires_grad_(True) for t in (x, w, b, add))
    yr = TF.conv2d(TF.interpolate(xr, scale_factor=2, mode="nearest"), wr, br, padding=1) + ar * 1.0
    gy = rn(*yr.shape, seed=3)
    yr.backward(gy)
    xg, wg, bg, ag = (dev(t).requires_grad_(True) for t in (x, w, b, add))
    yg = F.conv2d(xg, wg, bg, 1, 1, up=1, addend=ag * 1.0)
    yg.backward(dev(gy))
    for n, a, r in (("y", yg, yr), ("dx", xg.grad, xr.grad), ("dw", wg.grad, wr.grad), ("db", bg.grad, br.grad), ("dadd", ag.grad, ar.grad)):
        close(a, r, 1e-4, "up+addend " + n)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = TF.relu(TF.conv2d(TF.relu(xr), wr, br, padding=1))
    gy = rn(*yr.shape, seed=4)
    yr.backward(gy)
    xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
    yg = F.conv2d(xg, wg, bg, 1, 1, in_relu=True, relu=True)
    yg.backward(dev(gy))
    for n, a, r in (("y", yg, yr), ("dx", xg.grad, xr.grad), ("dw", wg.grad, wr.grad), ("db", bg.grad, br.grad)):
        close(a, r, 1e-4, "in_relu+relu " + n)


@pytest.mark.parametrize("shape", [(3, 16, 8, 8, 24), (2, 128, 16, 16, 64), (1, 5, 3, 4, 7)])
def test_conv_transpose(shape):
    from agl import functional as F
    N, Cin, H, W, Cout = shape
    x, w = rn(N, Cin, H, W), rn(Cin, Cout, 4, 4, seed=1) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = TF.conv_transpose2d(xr, wr, None, stride=2, padding=1)
    gy = rn(*yr.shape, seed=3)
    yr.backward(gy)
    xg, wg = dev(x).requires_grad_(True), dev(w).requires_grad_(True)
    yg = F.conv_transpose2d_k4s2p1(xg, wg)
    yg.backward(dev(gy))
    close(yg, yr, 2e-5, "y")
    close(xg.grad, xr.grad, 1e-4, "dx")
    close(wg.grad, wr.grad, 1e-4, "dw")


@pytest.mark.parametrize("shape", [(7, 6, 5, 4), (24, 64, 33, 33), (9, 1024, 2, 2), (50, 128, 1, 1), (3, 16, 40, 40)])
@pytest.mark.parametrize("mode", ["plain", "affine", "cond", "spade"])
def test_norm_modes(shape, mode):
    from agl import functional as F
    N, Cc, H, W = shape
    x = rn(N, Cc, H, W) * 1.7 + 0.4
    gy = rn(N, Cc, H, W, seed=9)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    nbt = torch.zeros((), dtype=torch.long)
    xr = x.clone().requires_grad_(True)
    xg = dev(x).requires_grad_(True)
    rmg, rvg, nbtg = dev(rm.clone()), dev(rv.clone()), dev(nbt.clone())
    extra = []
    if mode in ("plain", "affine"):
        wt, bs = (rn(Cc, seed=4) * 0.3 + 1, rn(Cc, seed=5) * 0.2) if mode == "affine" else (None, None)
        wr_ = wt.clone().requires_grad_(True) if wt is not None else None
        br_ = bs.clone().requires_grad_(True) if bs is not None else None
        res = rn(N, Cc, H, W, seed=6)
        yr = TF.relu(TF.batch_norm(xr, rm, rv, wr_, br_, True, 0.1, 1e-5)) if mode == "plain" else TF.batch_norm(xr, rm, rv, wr_, br_, True, 0.1, 1e-5) + res
        wg_ = dev(wt).requires_grad_(True) if wt is not None else None
        bg_ = dev(bs).requires_grad_(True) if bs is not None else None
        yg = F.batch_norm(xg, rmg, rvg, nbtg, wg_, bg_, relu=(mode == "plain"), residual=dev(res) if mode == "affine" else None)
        if wt is not None:
            extra = [("dweight", wg_, wr_), ("dbias", bg_, br_)]
    elif mode == "cond":
        V = 5
        table = torch.cat([1 + 0.1 * rn(V, Cc, seed=4), 0.1 * rn(V, Cc, seed=5)], dim=1)
        labels = torch.randint(0, V, (N,), generator=torch.Generator().manual_seed(3))
        tr = table.clone().requires_grad_(True)
        gb = TF.embedding(labels, tr)
        yr = TF.relu(gb[:, :Cc, None, None] * TF.batch_norm(xr, rm, rv, None, None, True, 0.1, 1e-5) + gb[:, Cc:, None, None])
        tg = dev(table).requires_grad_(True)
        yg = F.cond_batch_norm(xg, tg, dev(labels), rmg, rvg, nbtg, relu=True)
        extra = [("dtable", tg, tr)]
    else:
        gbt = rn(N, 2 * Cc, H, W, seed=4) * 0.5
        gr = gbt.clone().requires_grad_(True)
        yr = TF.relu(TF.batch_norm(xr, rm, rv, None, None, True, 0.1, 1e-5) * (1 + gr[:, :Cc]) + gr[:, Cc:])
        gg = dev(gbt).requires_grad_(True)
        yg = F.spade_modulate(xg, gg, rmg, rvg, nbtg, relu=True)
        extra = [("dgb", gg, gr)]
    yr.backward(gy)
    yg.backward(dev(gy))
    close(yg, yr, 2e-5, "y")
    close(xg.grad, xr.grad, 2e-4, "dx")
    for n, a, r in extra:
        close(a.grad, r.grad, 2e-4, n)
    close(rmg, rm, 2e-5, "running_mean")
    close(rvg, rv, 2e-5, "running_var")
    assert int(nbtg) == 1


def test_crop_matches_reference_fixture(golden_dir):
    from agl import functional as F
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    feats, boxes = torch.from_numpy(g["crop_feats"]), torch.from_numpy(g["crop_boxes"])
    for tag in ("sorted", "unsorted"):
        o2i = torch.from_numpy(g[f"crop_{tag}_o2i"])
        for HH, WW in ((8, 8), (5, 7), (32, 32)):
            k = f"crop_{tag}_{HH}x{WW}"
            fg = dev(feats).requires_grad_(True)
            y = F.crop_boxes(fg, dev(boxes), dev(o2i), HH, WW)
            y.backward(dev(torch.from_numpy(g[k + "_gy"])))
            close(y, torch.from_numpy(g[k + "_y"]), 2e-5, k)
            close(fg.grad, torch.from_numpy(g[k + "_dfeats"]), 1e-4, k + " dfeats")


def test_crop_backward_in_fixed_order_for_a_sorted_box_map(golden_dir):
    """agl_crop_bwd_sorted: with a non-decreasing box -> image map the gradient is a gather in fixed order —
    equal to the reference fixture, to torch's grid_sample backward on boxes that leave the map / are tiny / degenerate, to the
    scatter form, and bit-identical from run to run (the scatter's float atomics are not)."""
    from agl import functional as F
    from agl import lib as L
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    feats, boxes, o2i = torch.from_numpy(g["crop_feats"]), torch.from_numpy(g["crop_boxes"]), torch.from_numpy(g["crop_sorted_o2i"])
    od = L.box_map_to_device(o2i, DEV)
    assert od._agl_sorted and not getattr(L.box_map_to_device(torch.from_numpy(g["crop_unsorted_o2i"]), DEV), "_agl_sorted", False)
    for HH, WW in ((8, 8), (5, 7), (32, 32)):
        k = f"crop_sorted_{HH}x{WW}"
        fg = dev(feats).requires_grad_(True)
        F.crop_boxes(fg, dev(boxes), od, HH, WW).backward(dev(torch.from_numpy(g[k + "_gy"])))
        close(fg.grad, torch.from_numpy(g[k + "_dfeats"]), 1e-4, k + " dfeats (gather form)")
    # boxes past the borders, a one-pixel box, a zero-width box, a flipped box, several boxes per image, an image without boxes
    N, Cc, H, W, s = 5, 3, 24, 40, 16
    bx = torch.tensor([[-0.2, -0.1, 0.5, 0.6], [0.3, 0.2, 1.3, 1.1], [0.5, 0.5, 0.52, 0.53], [0.4, 0.1, 0.4, 0.9], [0.9, 0.8, 0.1, 0.2],
                       [0.0, 0.0, 1.0, 1.0], [0.1, 0.6, 0.35, 0.95], [0.25, 0.25, 0.75, 0.75]])
    o2 = torch.tensor([0, 0, 0, 1, 1, 3, 4, 4])
    x = rn(N, Cc, H, W)
    gy = rn(bx.shape[0], Cc, s, s, seed=1)
    xr = x.clone().requires_grad_(True)
    t = torch.linspace(0, 1, s)
    X = (1 - t).view(1, 1, s) * (2 * bx[:, 0] - 1).view(-1, 1, 1) + t.view(1, 1, s) * (2 * bx[:, 2] - 1).view(-1, 1, 1)
    Y = (1 - t).view(1, s, 1) * (2 * bx[:, 1] - 1).view(-1, 1, 1) + t.view(1, s, 1) * (2 * bx[:, 3] - 1).view(-1, 1, 1)
    grid = torch.stack([X.expand(-1, s, s), Y.expand(-1, s, s)], dim=3)
    TF.grid_sample(xr[o2], grid, mode="bilinear", padding_mode="zeros", align_corners=False).backward(gy)
    d_sorted = L.crop_bwd(dev(gy), dev(bx), L.box_map_to_device(o2, DEV), (N, Cc, H, W))
    d_atomic = L.crop_bwd(dev(gy), dev(bx), dev(o2), (N, Cc, H, W))
    close(d_sorted, xr.grad, 2e-5, "gather form vs torch grid_sample backward")
    close(d_sorted, d_atomic.cpu(), 2e-6, "gather form vs scatter form")
    assert float(d_sorted[2].abs().max()) == 0.0, "an image without boxes receives no gradient"
    for _ in range(3):
        assert torch.equal(L.crop_bwd(dev(gy), dev(bx), L.box_map_to_device(o2, DEV), (N, Cc, H, W)), d_sorted), "not reproducible"


def test_pool_upsample_sum_reparam_maskouter():
    from agl import functional as F
    x = rn(3, 5, 8, 12)
    for in_relu in (False, True):
        xr, xg = x.clone().requires_grad_(True), dev(x).requires_grad_(True)
        yr = TF.avg_pool2d(TF.relu(xr) if in_relu else xr, 2)
        gy = rn(*yr.shape, seed=1)
        yr.backward(gy)
        yg = F.avg_pool2(xg, in_relu)
        yg.backward(dev(gy))
        close(yg, yr, 1e-6, "avgpool")
        close(xg.grad, xr.grad, 1e-6, "avgpool dx")
        xr, xg = x.clone().requires_grad_(True), dev(x).requires_grad_(True)
        yr = (TF.relu(xr) if in_relu else xr).sum(dim=(2, 3)) * 0.5
        gy = rn(*yr.shape, seed=2)
        yr.backward(gy)
        yg = F.sum_hw(xg, in_relu, 0.5)
        yg.backward(dev(gy))
        close(yg, yr, 1e-5, "sum_hw")
        close(xg.grad, xr.grad, 1e-6, "sum_hw dx")
    for k in (1, 3):
        xr, xg = x.clone().requires_grad_(True), dev(x).requires_grad_(True)
        yr = TF.interpolate(xr, scale_factor=2 ** k, mode="nearest")
        gy = rn(*yr.shape, seed=3)
        yr.backward(gy)
        yg = F.upsample_nearest(xg, k)
        yg.backward(dev(gy))
        close(yg, yr, 0, "upsample")
        close(xg.grad, xr.grad, 1e-5, "upsample dx")
    mu, lv, eps = rn(6, 9), rn(6, 9, seed=1), rn(6, 9, seed=2)
    mr, lr_ = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    zr = eps * torch.exp(0.5 * lr_) + mr
    gz = rn(6, 9, seed=3)
    zr.backward(gz)
    mg, lg = dev(mu).requires_grad_(True), dev(lv).requires_grad_(True)
    zg = F.reparameterize(mg, lg, dev(eps))
    zg.backward(dev(gz))
    close(zg, zr, 1e-6, "z"); close(mg.grad, mr.grad, 1e-6, "dmu"); close(lg.grad, lr_.grad, 1e-5, "dlogvar")
    u, mask = rn(4, 6), (rn(4, 1, 10, 10, seed=1) > 0).float()
    ur = u.clone().requires_grad_(True)
    yr = TF.pad(ur[:, :, None, None] * mask, (1, 1, 1, 1))
    gy = rn(*yr.shape, seed=2)
    yr.backward(gy)
    ug = dev(u).requires_grad_(True)
    yg = F.mask_outer(ug, dev(mask), 1)
    yg.backward(dev(gy))
    close(yg, yr, 0, "mask_outer"); close(ug.grad, ur.grad, 1e-5, "mask_outer du")


def test_spectral_norm_weights():
    from agl import functional as F
    import oracle.graph as OG
    shapes = [(64, 3, 3, 3), (128, 64, 3, 3), (20, 7, 1, 1), (1, 1024), (179, 1024), (300, 200, 3, 3)]
    P = {}
    for i, s in enumerate(shapes):
        P[f"l{i}.weight_orig"] = (rn(*s, seed=i) * 0.1).requires_grad_(True)
        P[f"l{i}.weight_u"] = TF.normalize(rn(s[0], seed=10 + i), dim=0)
        P[f"l{i}.weight_v"] = TF.normalize(rn(int(np.prod(s[1:])), seed=20 + i), dim=0)
    ws = [dev(P[f"l{i}.weight_orig"].detach()).requires_grad_(True) for i in range(len(shapes))]
    us = [dev(P[f"l{i}.weight_u"].clone()) for i in range(len(shapes))]
    vs = [dev(P[f"l{i}.weight_v"].clone()) for i in range(len(shapes))]
    for it in range(3):
        outs_g = F.spectral_norm_weights(ws, us, vs, True)
        outs_r = [OG.sn_weight(P, f"l{i}.", True) for i in range(len(shapes))]
        for i in range(len(shapes)):
            close(outs_g[i], outs_r[i], 2e-5, f"w_sn[{i}] iter {it}")
            close(us[i], P[f"l{i}.weight_u"], 2e-5, f"u[{i}] iter {it}")
            close(vs[i], P[f"l{i}.weight_v"], 2e-5, f"v[{i}] iter {it}")
    gs = [rn(*s, seed=30 + i) for i, s in enumerate(shapes)]
    sum((o * g).sum() for o, g in zip(outs_r, gs)).backward()
    torch.autograd.backward(outs_g, [dev(g) for g in gs])
    for i in range(len(shapes)):
        close(ws[i].grad, P[f"l{i}.weight_orig"].grad, 1e-4, f"dw_orig[{i}]")
    # eval mode: no power iteration, u/v untouched
    u_before = us[0].clone()
    outs_e = F.spectral_norm_weights(ws, us, vs, False)
    close(outs_e[0], OG.sn_weight(P, "l0.", False), 2e-5, "eval w_sn")
    assert torch.equal(u_before, us[0])


def test_convlstm_matches_reference_fixture(golden_dir):
    from agl.generator import LayoutConvLSTM
    g = np.load(os.path.join(golden_dir, "ops_small.npz"))
    from oracle.fill import fill_state
    m = LayoutConvLSTM(8, 12, [8, 4, 4], (5, 5))
    m.load_state_dict(fill_state(m.state_dict()))
    m = m.to(DEV)
    x = dev(torch.from_numpy(g["clstm_x"])).requires_grad_(True)
    y = m(x, torch.from_numpy(g["clstm_o2i"]))
    y.backward(dev(torch.from_numpy(g["clstm_gy"])))
    close(y, torch.from_numpy(g["clstm_y"]), 2e-5, "convlstm y")
    close(x.grad, torch.from_numpy(g["clstm_dx"]), 2e-4, "convlstm dx")
    close(m.cell_list[0].conv.weight.grad, torch.from_numpy(g["clstm_dW0"]), 2e-4, "convlstm dW0")
    close(m.cell_list[2].conv.weight.grad, torch.from_numpy(g["clstm_dW2"]), 2e-4, "convlstm dW2")
    close(m.cell_list[2].conv.bias.grad, torch.from_numpy(g["clstm_db2"]), 2e-4, "convlstm db2")


@pytest.mark.parametrize("mode", ["f32", "split3", "bf16"])
@pytest.mark.parametrize("case", [(393, 1024, 179), (64, 1024, 1), (393, 64, 2048), (5, 100, 7), (37, 106, 64), (9, 128, 130)])
def test_linear_layers_forward_and_gradients(case, mode):
    """nn.Linear as the path has it (discriminator heads discriminator.py:139-141, :181-185, :226; the crop encoder's fc layers,
    generator_obj_att.py:418-422; the attribute encoder :590-600) — agl_conv2d_* on 1x1 maps: the forward on the generic exact GEMM, the
    two gradients on the dedicated kernels of csrc/few.hip (linear_bww_k, linear_bwd_data_k).  Forward with bias / input ReLU / output
    ReLU / accumulation, input and weight gradients, against torch: exact arithmetic to 1e-5 in 'f32' and 'split3' (both run the
    exact kernels), in 'bf16' against the fp32 product of the bf16-rounded operands."""
    from agl import lib as L
    N, Cin, Cout = case
    x, w, b = rn(N, Cin, 1, 1), rn(Cout, Cin, 1, 1, seed=1) * (1.0 / Cin ** 0.5), rn(Cout, seed=2)
    gy = rn(N, Cout, 1, 1, seed=3)
    flags = {"f32": 0, "split3": L.CONV_SPLIT3, "bf16": L.CONV_BF16}[mode]
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mode == "bf16" else (lambda t: t)
    xr, wr = r(TF.relu(x)).requires_grad_(True), r(w).requires_grad_(True)
    yr = TF.relu(TF.conv2d(xr, wr, b))
    with L.conv_flags(flags):
        y = L.conv2d_fwd(dev(x), dev(w), dev(b), 1, 0, in_relu=True, relu=True)
        base = rn(N, Cout, 1, 1, seed=4)
        y2 = L.conv2d_fwd(dev(x), dev(w), None, 1, 0, out=dev(base).clone(), accumulate=True)
        dx = L.conv2d_bwd_data(dev(gy), dev(w), (1, 1), 1, 0)
        dw = L.conv2d_bwd_weight(dev(gy), dev(x), 1, 1, 0)
    close(y, yr, 1e-5, "linear forward (bias, input ReLU, output ReLU)")
    close(y2, base + TF.conv2d(r(x), r(w)), 1e-5, "linear forward, accumulating")
    xg, wg = r(x).requires_grad_(True), r(w).requires_grad_(True)
    TF.conv2d(xg, wg).backward(r(gy))
    close(dx, xg.grad, 1e-5, "linear input gradient")
    close(dw, wg.grad, 2e-5, "linear weight gradient")


def test_row_losses_over_several_workgroups_equal_the_one_workgroup_kernels():
    """agl_cross_entropy_ws / agl_bce_logits_posw_ws (16 rows per workgroup, the row terms added by one workgroup in a fixed order)
    against the single-workgroup entries they replace in the training loop (train64.py:241-245, :323-354): gradients bit-identical,
    the cross-entropy value bit-identical, the attribute loss's value to double rounding; un-annotated rows and a bad label as there."""
    from agl import lib as L
    from agl import losses as LS
    g = torch.Generator().manual_seed(3)
    rows, A, V = 393, 106, 179
    x = torch.randn(rows, A, generator=g)
    t = (torch.rand(rows, A, generator=g) < 0.03).float()
    t[::4] = 0
    pw = torch.rand(A, generator=g) * 20 + 1
    lg = torch.randn(rows, V, generator=g) * 3
    lab = torch.randint(0, V, (rows,), generator=g)
    xd, td, pd, ld, labd = dev(x), dev(t), dev(pw), dev(lg), dev(lab)
    s_new, s_old = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    dx_new = LS.bce_posw(xd, td, pd, 0.7, s_new)
    dx_old = torch.empty_like(xd)
    L.call("agl_bce_logits_posw", L.ptr(xd), L.ptr(td), L.ptr(pd), rows, A, 0.7, L.ptr(s_old), L.ptr(dx_old), L.stream())
    assert torch.equal(dx_new, dx_old)
    assert abs(float(s_new) - float(s_old)) <= 1e-6 * abs(float(s_old))
    dl_new = LS.cross_entropy(ld, labd, 1.3, s_new)
    dl_old = torch.empty_like(ld)
    L.call("agl_cross_entropy", L.ptr(ld), L.ptr(labd, torch.int64), rows, V, 1.3, L.ptr(s_old), L.ptr(dl_old), L.stream())
    assert torch.equal(dl_new, dl_old) and float(s_new) == float(s_old)
    labd[7] = V      # a label outside [0, V): the loss and that gradient row are poisoned, the other rows are not
    dl_bad = LS.cross_entropy(ld, labd, 1.3, s_new)
    assert torch.isnan(s_new).all() and torch.isnan(dl_bad[7]).all() and torch.equal(dl_bad[8:], dl_old[8:])
    td.zero_()       # no annotated row at all: NaN loss (0 / 0 in the reference), zero gradient
    dx0 = LS.bce_posw(xd, td, pd, 0.7, s_new)
    assert torch.isnan(s_new).all() and float(dx0.abs().max()) == 0.0


@pytest.mark.parametrize("hw", [2, 4])
def test_sum_over_small_maps_one_row_per_thread(hw):
    """agl_sum_hw_fwd on the discriminators' last maps (discriminator.py:139, :181, :226: sum over 2x2 / 4x4 after the ReLU): one row
    per thread instead of one per wave.  Against torch, and BIT-identical to the wave-per-row kernel (which a tensor that is not
    16-byte aligned still takes): the pieces are added in the wave reduction's order."""
    from agl import lib as L
    N, Cc = 37, 1024
    x = rn(N, Cc, hw, hw)
    for in_relu in (False, True):
        ref = (TF.relu(x) if in_relu else x).sum(dim=(2, 3)) * 0.5
        a = L.sum_hw_fwd(dev(x), in_relu, 0.5)
        buf = torch.empty(x.numel() + 1, dtype=torch.float32, device=DEV)
        xm = buf[1:].view(N, Cc, hw, hw)          # 4 bytes off a 16-byte boundary: the wave-per-row kernel
        xm.copy_(dev(x))
        assert xm.data_ptr() % 16 != 0 and xm.is_contiguous()
        b = L.sum_hw_fwd(xm, in_relu, 0.5)
        close(a, ref, 1e-5, "sum over the map")
        assert torch.equal(a, b), float((a - b).abs().max())


@pytest.mark.parametrize("mode", ["f32", "split3", "bf16"])
def test_convlstm_gate_kernels_adding_the_recurrence_convolutions_partial_sums(mode):
    """AGL_CONV_DEFER_SUM (include/agl.h): the recurrence convolutions of LayoutConvLSTM (generator_obj_att.py:99-104, :306-331) cut
    their reduction over workgroups on the few images still active at a step; with the flag the partial outputs stay in the workspace
    and the gate kernel that follows adds them (agl_lstm_gates_fwd_sum / _bwd_sum) in the epilogue's own order.  At the extents of the
    64 px model (512 -> 128 -> 64 -> 64 hidden channels on 8x8 maps, sequences of 1..7 objects): output and every gradient
    BIT-identical to the schedule with the epilogue launches, in all three arithmetics, and at least one launch less per
    recurrence step overall."""
    from agl import convlstm as CL
    from agl import lib as L
    from agl.generator import LayoutConvLSTM
    torch.manual_seed(5)
    m = LayoutConvLSTM(8, 512, [128, 64, 64], (5, 5)).to(DEV)
    lens = [7, 3, 5, 1, 4, 6, 2, 5, 3]
    o2i = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(lens)])
    x = rn(int(o2i.numel()), 512, 8, 8)
    gy = rn(len(lens), 64, 8, 8, seed=9)
    flags = {"f32": 0, "split3": L.CONV_SPLIT3, "bf16": L.CONV_BF16}[mode]
    res, deferred = [], []
    seen = []
    orig = L.DeferredSum.__init__

    def spy(self, out, ws):
        orig(self, out, ws)
        seen.append(self.splits)
    for on in (True, False):
        prev, CL.DEFER_SUM = CL.DEFER_SUM, on
        L.DeferredSum.__init__ = spy
        del seen[:]
        try:
            m.zero_grad(set_to_none=True)
            xd = dev(x).requires_grad_(True)
            with L.conv_flags(flags):
                y = m(xd, o2i)
                y.backward(dev(gy))
            torch.cuda.synchronize()
        finally:
            CL.DEFER_SUM = prev
            L.DeferredSum.__init__ = orig
        deferred.append(sum(1 for v in seen if v >= 2))
        res.append([y.detach().clone(), xd.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    assert deferred[1] == 0 and deferred[0] >= 6, ("recurrence convolutions that left their slabs to the gate kernel", deferred)
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), ("tensor", i, float((a - b).abs().max()))


@pytest.mark.parametrize("case", [CONV_CASES[1], CONV_CASES[2], CONV_CASES[3], CONV_CASES[4], CONV_CASES[8]])
def test_conv2d_bf16_operand_mode(case):
    """bf16 MFMA mode (BASELINE configs 3/5): operands are the RNE bf16 roundings of the fp32 tensors, accumulation is
    fp32 — so the result must equal an fp32 convolution of the rounded tensors to fp32 accuracy, and the plain fp32
    result to bf16 accuracy (2^-8 per operand)."""
    from agl import functional as F, lib as L
    N, Cin, H, W, Cout, ks, s, p = case
    x, w = rn(N, Cin, H, W), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5)
    r = lambda t: t.to(torch.bfloat16).to(torch.float32)
    xr, wr = r(x).requires_grad_(True), r(w).requires_grad_(True)
    yr = TF.conv2d(xr, wr, None, stride=s, padding=p)
    gy = rn(*yr.shape, seed=3)
    yr.backward(r(gy))                       # reference of the backward passes: rounded dy as well
    y32 = TF.conv2d(x, w, None, stride=s, padding=p)
    with L.conv_flags(L.CONV_BF16):
        xg, wg = dev(x).requires_grad_(True), dev(w).requires_grad_(True)
        yg = F.conv2d(xg, wg, None, s, p)
        yg.backward(dev(gy))
    close(yg, yr, 2e-5, "y vs fp32 conv of bf16-rounded operands")
    close(yg, y32, 2e-2, "y vs plain fp32")
    close(xg.grad, xr.grad, 1e-4, "dx")
    close(wg.grad, TF.conv2d(r(x).transpose(0, 1), r(gy).transpose(0, 1), stride=1, padding=p, dilation=s).transpose(0, 1)[:, :, :ks, :ks]
          if False else wr.grad, 1e-2, "dw")        # dw uses rounded x and rounded dy: compare at bf16 accuracy


PATCH_CASES = [
    # N, Cin, H, W, Cout, ks   (stride 1, "same" padding) — shapes routed to the LDS-patch kernel
    (2, 64, 16, 16, 128, 3), (3, 16, 32, 16, 64, 3), (5, 8, 4, 4, 64, 3), (9, 24, 4, 4, 136, 3), (3, 64, 8, 8, 512, 5),
    (3, 40, 8, 8, 200, 3), (2, 6, 16, 32, 96, 5), (17, 10, 4, 4, 48, 5), (1, 128, 64, 64, 64, 3),
]


@pytest.mark.parametrize("case", PATCH_CASES)
def test_patch_conv_fwd_bwd(case):
    from agl import functional as F
    N, Cin, H, W, Cout, ks = case
    p = ks // 2
    x, w, b = rn(N, Cin, H, W), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = TF.relu(TF.conv2d(TF.relu(xr), wr, br, padding=p))
    gy = rn(*yr.shape, seed=3)
    yr.backward(gy)
    xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
    yg = F.conv2d(xg, wg, bg, 1, p, in_relu=True, relu=True)
    yg.backward(dev(gy))
    close(yg, yr, 2e-5, "y")
    close(xg.grad, xr.grad, 1e-4, "dx")
    close(wg.grad, wr.grad, 1e-4, "dw")
    close(bg.grad, br.grad, 1e-4, "db")


def test_patch_conv_upsampled_input_and_addend():
    from agl import functional as F
    x, w, b = rn(3, 64, 8, 8), rn(128, 64, 3, 3, seed=1) * 0.05, rn(128, seed=2)
    add = rn(3, 128, 32, 32, seed=5)
    xr, wr, br, ar = (t.clone().requires_grad_(True) for t in (x, w, b, add))
    yr = TF.conv2d(TF.interpolate(xr, scale_factor=4, mode="nearest"), wr, br, padding=1) + ar * 1.0
    gy = rn(*yr.shape, seed=3)
    yr.backward(gy)
    xg, wg, bg, ag = (dev(t).requires_grad_(True) for t in (x, w, b, add))
    yg = F.conv2d(xg, wg, bg, 1, 1, up=2, addend=ag * 1.0)
    yg.backward(dev(gy))
    for n, a, r in (("y", yg, yr), ("dx", xg.grad, xr.grad), ("dw", wg.grad, wr.grad), ("db", bg.grad, br.grad), ("dadd", ag.grad, ar.grad)):
        close(a, r, 1e-4, "patch up+addend " + n)


@pytest.mark.parametrize("shape", [(393, 64, 66, 128, 4, 2, 1), (393, 512, 8, 512, 5, 1, 2), (64, 128, 64, 128, 3, 1, 1), (393, 256, 8, 512, 3, 1, 1),
                                   (64, 64, 64, 3, 7, 1, 3), (32, 128, 128, 3, 7, 1, 3), (393, 3, 32, 64, 7, 1, 3)])
def test_conv_adjoint_identities_at_full_size(shape):
    """Oracle-free properties at BASELINE config-2 sizes (batch 64 / 393 objects): the three convolution passes are
    mutually adjoint — <conv(x,w), g> = <x, bwd_data(g,w)> = <w, bwd_weight(g,x)> — and the forward pass is linear."""
    from agl import lib as L
    N, Cin, H, Cout, ks, s, p = shape
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, Cin, H, H, device=DEV, generator=gen)
    w = torch.randn(Cout, Cin, ks, ks, device=DEV, generator=gen) * (1.0 / (Cin * ks * ks) ** 0.5)
    y = L.conv2d_fwd(x, w, None, s, p)
    g = torch.randn(y.shape, device=DEV, generator=gen)
    dx = L.conv2d_bwd_data(g, w, (H, H), s, p)
    dw = L.conv2d_bwd_weight(g, x, ks, s, p)
    a = float((y.double() * g.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * dw.double()).sum())
    scale = float(y.double().norm() * g.double().norm())
    assert abs(a - b) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, b, c, scale)
    x2 = torch.randn(N, Cin, H, H, device=DEV, generator=gen)
    lin = L.conv2d_fwd(0.5 * x - 2.0 * x2, w, None, s, p)
    close(lin, 0.5 * y - 2.0 * L.conv2d_fwd(x2, w, None, s, p), 2e-5, "linearity")


POS_CASES = [
    # N, Cin, H, Cout, ks, stride, pad        (position-major path: maps <= 8x8, >= 96 images, Cin a power of two)
    (130, 64, 8, 96, 5, 1, 2),      # ConvLSTM-like
    (130, 64, 8, 128, 5, 1, 2),
    (393, 128, 8, 256, 5, 1, 2),
    (100, 64, 6, 64, 5, 1, 2),      # ragged 6x6 map
    (393, 128, 8, 64, 3, 1, 1),
    (100, 256, 4, 128, 3, 1, 1),
    (200, 64, 8, 128, 4, 2, 1),     # 8 -> 4
    (129, 128, 4, 256, 4, 2, 1),    # 4 -> 2
    (97, 64, 6, 64, 3, 1, 1),       # ragged map
]


@pytest.mark.parametrize("case", POS_CASES)
def test_position_major_conv_vs_torch_and_im2col(case):
    """Small-map convolutions run as per-position sums of plain matrix products (padded taps are never visited);
    results must match torch and the im2col/patch kernels (flag AGL_CONV_NO_POS) to fp32 rounding, including the
    fused input ReLU, bias, output ReLU and accumulate epilogues."""
    from agl import lib as L
    N, Cin, H, Cout, ks, s, p = case
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    yr = TF.conv2d(torch.relu(x), w, b, stride=s, padding=p)
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = L.conv2d_fwd(xd, wd, bd, s, p, in_relu=True)
    close(y, yr, 2e-5, "y (input ReLU, bias)")
    base = rn(*yr.shape, seed=5)
    y2 = L.conv2d_fwd(xd, wd, None, s, p, out=dev(base).clone(), accumulate=True)
    close(y2, base + TF.conv2d(x, w, None, stride=s, padding=p), 2e-5, "accumulate")
    y3 = L.conv2d_fwd(xd, wd, bd, s, p, relu=True)
    close(y3, torch.relu(TF.conv2d(x, w, b, stride=s, padding=p)), 2e-5, "output ReLU")
    with L.conv_flags(L.CONV_NO_POS):
        y_ref = L.conv2d_fwd(xd, wd, bd, s, p, in_relu=True)
    close(y, y_ref, 5e-6, "position-major vs im2col")
    # input gradient (same path with flipped taps, fused positive mask) and weight gradient (position-major reduction)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yy = TF.conv2d(xr, wr, None, stride=s, padding=p)
    g = rn(*yy.shape, seed=7)
    yy.backward(g)
    gd = dev(g)
    dx = L.conv2d_bwd_data(gd, wd, (H, H), s, p)
    close(dx, xr.grad, 1e-4, "dx")
    dxm = L.conv2d_bwd_data(gd, wd, (H, H), s, p, pos_mask=xd)
    close(dxm, xr.grad * (x > 0), 1e-4, "dx masked")
    dw = L.conv2d_bwd_weight(gd, xd, ks, s, p)
    close(dw, wr.grad, 1e-4, "dw")
    base_w = rn(*w.shape, seed=9)
    dw2 = L.conv2d_bwd_weight(gd, xd, ks, s, p, out=dev(base_w).clone(), accumulate=True)
    close(dw2, base_w + wr.grad, 1e-4, "dw accumulate")


def test_relu_gradient_masked_by_consumer():
    """conv(relu=True, relu_grad_by_consumer=True) -> conv(x_relu=True): the consumer masks its input gradient by
    x > 0 in the input-gradient epilogue, so the producer skips the ReLU-backward pass; gradients must equal the plain
    composition (torch reference), for the im2col, patch and fused 4x4/stride-2 consumers."""
    from agl import functional as F
    for (N, C0, H, C1, C2, fused_pool) in ((3, 16, 16, 64, 48, False), (2, 8, 8, 32, 24, True), (5, 64, 32, 64, 128, True)):
        x, w1, b1 = rn(N, C0, H, H), rn(C1, C0, 3, 3, seed=1) * 0.1, rn(C1, seed=2)
        w2, b2 = rn(C2, C1, 3, 3, seed=3) * 0.1, rn(C2, seed=4)
        ts = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        h = TF.relu(TF.conv2d(ts[0], ts[1], ts[2], padding=1))
        y = TF.conv2d(h, ts[3], ts[4], padding=1)
        if fused_pool:
            y = TF.avg_pool2d(y, 2)
        gy = rn(*y.shape, seed=5)
        y.backward(gy)
        tg = [dev(t).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        hg = F.conv2d(tg[0], tg[1], tg[2], 1, 1, relu=True, relu_grad_by_consumer=True)
        yg = F.conv3x3_avgpool2(hg, tg[3], tg[4], x_relu=True) if fused_pool else F.conv2d(hg, tg[3], tg[4], 1, 1, x_relu=True)
        close(yg, y, 3e-5, "y")
        yg.backward(dev(gy))
        for nm, a, r in zip(("dx", "dw1", "db1", "dw2", "db2"), tg, ts):
            close(a.grad, r.grad, 2e-4, nm)


@pytest.mark.parametrize("shape", [(3, 16, 16, 24), (2, 64, 32, 128), (5, 128, 8, 256), (4, 256, 4, 64), (130, 64, 2, 64)])
def test_conv3x3_avgpool_box_form(shape):
    """avg_pool2(conv3x3(x)) as a 3x3 stride-2 convolution of the box-filtered, zero-extended input: forward and all
    gradients against torch, and against the pooled-filter (4x4 stride-2) form."""
    from agl import functional as F
    N, Cin, H, Cout = shape
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, 3, 3, seed=1) * (1.0 / (Cin * 9) ** 0.5), rn(Cout, seed=2)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = TF.avg_pool2d(TF.conv2d(xr, wr, br, padding=1), 2)
    gy = rn(*yr.shape, seed=3)
    yr.backward(gy)
    outs = []
    for box in (True, False):
        F.BOX_FORM = box
        try:
            xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
            yg = F.conv3x3_avgpool2(xg, wg, bg)
            yg.backward(dev(gy))
        finally:
            F.BOX_FORM = True
        close(yg, yr, 3e-5, f"y (box={box})")
        close(xg.grad, xr.grad, 1e-4, f"dx (box={box})")
        close(wg.grad, wr.grad, 1e-4, f"dw (box={box})")
        close(bg.grad, br.grad, 1e-4, f"db (box={box})")
        outs.append(yg.detach())
    close(outs[0], outs[1], 2e-5, "box form vs pooled-filter form")


@pytest.mark.parametrize("case", [(5, 3, 32, 32, 64, 3), (5, 3, 32, 32, 64, 1), (2, 3, 64, 64, 64, 3), (3, 4, 16, 24, 48, 3), (9, 1, 8, 64, 16, 1),
                                  (2, 2, 40, 28, 130, 3)])
def test_few_input_channel_forward_stream(case):
    """Forward with <= 4 input channels, 1x1 / 3x3 (csrc/few.hip few_cin_fwd_k: the discriminators' first convolution and the 1x1
    shortcut accumulated onto the residual branch): bias, fused input ReLU, output ReLU and accumulation, borders included, against
    torch fp32 — exact fp32 arithmetic in every mode, so the same 2e-5 in all three."""
    from agl import lib as L
    N, Cin, H, W, Cout, ks = case
    x, w, b = rn(N, Cin, H, W), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    old = rn(N, Cout, H, W, seed=5)
    for flags in (0, L.CONV_SPLIT3, L.CONV_BF16):
        with L.conv_flags(flags):
            for in_relu, relu, acc in ((False, False, False), (True, True, False), (False, False, True), (True, False, True)):
                xin = torch.relu(x) if in_relu else x
                ref = TF.conv2d(xin, w, b, padding=ks // 2)
                if acc:
                    ref = ref + old
                if relu:
                    ref = torch.relu(ref)
                out = dev(old).clone() if acc else None
                y = L.conv2d_fwd(dev(x), dev(w), dev(b), 1, ks // 2, 0, in_relu, relu, out=out, accumulate=acc)
                close(y, ref, 2e-5, f"y flags={flags} in_relu={in_relu} relu={relu} acc={acc}")
    y = L.conv2d_fwd(dev(x), dev(w), None, 1, ks // 2)
    close(y, TF.conv2d(x, w, None, padding=ks // 2), 2e-5, "no bias")


@pytest.mark.parametrize("mode", ["f32", "split3", "bf16"])
@pytest.mark.parametrize("case", [(3, 64, 32, 64, True), (2, 3, 16, 32, False), (5, 128, 8, 128, True), (2, 48, 64, 80, True), (70, 64, 4, 64, True)])
def test_conv_and_pool_fork_of_a_block_input(case, mode):
    """x -> (conv3x3(x), avg_pool2(x)) as one graph node (agl.functional.conv2d_and_avg_pool2: the input of the discriminators'
    down-sampling blocks): values and every gradient equal the two separate functions' — the same kernels run, the input gradient
    is accumulated in the convolution's epilogue instead of by a separate addition, the same two fp32 terms either way."""
    from agl import functional as F, lib as L
    N, C, H, Co, in_relu = case
    flags = {"f32": 0, "split3": L.CONV_SPLIT3, "bf16": L.CONV_BF16}[mode]
    x, w, b = rn(N, C, H, H), rn(Co, C, 3, 3, seed=1) * (1.0 / (C * 9) ** 0.5), rn(Co, seed=2)
    gy, gs = rn(N, Co, H, H, seed=3), rn(N, C, H // 2, H // 2, seed=4)
    res = []
    with L.conv_flags(flags):
        for fork in (True, False):
            xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
            if fork:
                y, s = F.conv2d_and_avg_pool2(xg, wg, bg, 1, in_relu=in_relu, relu=True, relu_grad_by_consumer=False)
            else:
                y = F.conv2d(xg, wg, bg, 1, 1, in_relu=in_relu, relu=True)
                s = F.avg_pool2(xg, in_relu=in_relu)
            torch.autograd.backward([y, s], [dev(gy), dev(gs)])
            res.append((y.detach(), s.detach(), xg.grad, wg.grad, bg.grad))
    for a_, b_, what in zip(res[0], res[1], ("y", "s", "dx", "dw", "db")):
        assert torch.equal(a_, b_), f"{what} differs from the separate functions ({mode})"
    if mode == "f32":
        xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
        xin = torch.relu(xr) if in_relu else xr
        yr, sr = torch.relu(TF.conv2d(xin, wr, br, padding=1)), TF.avg_pool2d(xin, 2)
        torch.autograd.backward([yr, sr], [gy, gs])
        close(res[0][0], yr, 3e-5, "y")
        close(res[0][2], xr.grad, 1e-4, "dx")


@pytest.mark.parametrize("case", [(4, 64, 32, 128, 4, 1), (5, 32, 16, 64, 4, 1), (9, 16, 8, 96, 4, 1), (4, 64, 33, 128, 3, 0), (6, 32, 17, 64, 3, 0),
                                  (10, 24, 9, 72, 3, 0), (2, 128, 64, 64, 4, 1)])
def test_patch_conv_stride2_forward(case):
    """4x4/stride-2 and 3x3/stride-2 forward convolutions on the LDS-patch kernel (stride template): against torch and
    against the im2col kernel (flag AGL_CONV_NO_PATCH_S2 = stride-1 patches only), with bias / input ReLU / output ReLU."""
    from agl import lib as L
    N, Cin, H, Cout, ks, p = case
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    yr = TF.relu(TF.conv2d(TF.relu(x), w, b, stride=2, padding=p))
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = L.conv2d_fwd(xd, wd, bd, 2, p, in_relu=True, relu=True)
    close(y, yr, 2e-5, "y")
    with L.conv_flags(L.CONV_NO_PATCH_S2):
        y_ref = L.conv2d_fwd(xd, wd, bd, 2, p, in_relu=True, relu=True)
    close(y, y_ref, 5e-6, "patch vs im2col")


def test_grid_gather_and_spade_block_grids():
    """agl_grid_gather_fwd/bwd against torch indexing for every cached map kind, and SPADE with the two convolutions on
    the block-class grids (3 / 5 cells per constant block) against the same module on the full-resolution grid:
    outputs and all gradients must agree to fp32 rounding for f = 4 (3-class grid only), 8 and 16."""
    from agl import functional as F
    from agl.generator import SPADE
    for kind, blocks, f in (("up3", 8, 0), ("3to5", 8, 0), ("5tof", 8, 8), ("5tof", 4, 16), ("3tof", 8, 4)):
        m, lo, src = F._grid_map(kind, blocks, f, torch.device(DEV))
        x = rn(2, 3, src, src).to(DEV).requires_grad_(True)
        y = F.grid_gather(x, kind, blocks, f)
        mi = m.long()
        ref = x.detach()[:, :, mi][:, :, :, mi]
        assert torch.equal(y, ref), kind
        g = rn(*y.shape, seed=1).to(DEV)
        y.backward(g)
        xr = x.detach().clone().requires_grad_(True)
        xr[:, :, mi][:, :, :, mi].backward(g)
        close(x.grad, xr.grad, 1e-5, "grid_gather backward " + kind)
    for C, S in ((32, 32), (16, 64), (8, 128)):
        torch.manual_seed(C)
        sp = SPADE(C, 64).to(DEV)
        x0, seg0 = rn(2, C, S, S), rn(2, 64, 8, 8, seed=3)
        gy = rn(2, C, S, S, seed=4).to(DEV)
        res = []
        # block grids with the expansion folded into the modulation kernel's reads / block grids with the expanded gamma|beta
        # written out / both convolutions at full resolution: the first two must agree bit for bit in the forward
        for flag, fold in ((True, True), (True, False), (False, False)):
            sp.block_grids, sp.fold_gather = flag, fold
            sp.zero_grad()
            x, seg = dev(x0).requires_grad_(True), dev(seg0).requires_grad_(True)
            y = sp(x, seg, relu=True)
            y.backward(gy)
            res.append((y.detach(), x.grad, seg.grad, [p.grad.clone() for p in sp.parameters()]))
        (y1, dx1, ds1, gp1), (y1b, dx1b, ds1b, gp1b), (y2, dx2, ds2, gp2) = res
        assert torch.equal(y1, y1b), "folded expansion must read exactly what the written-out expansion holds"
        close(dx1, dx1b, 1e-6, f"SPADE dx folded vs written out S={S}")
        close(ds1, ds1b, 1e-5, f"SPADE dseg folded vs written out S={S}")
        for a, r in zip(gp1, gp1b):
            close(a, r, 1e-5, f"SPADE parameter gradient folded vs written out S={S}")
        close(y1, y2, 1e-5, f"SPADE y S={S}")
        close(dx1, dx2, 1e-5, f"SPADE dx S={S}")
        close(ds1, ds2, 2e-4, f"SPADE dseg S={S}")
        for a, r in zip(gp1, gp2):
            close(a, r, 2e-4, f"SPADE parameter gradient S={S}")


def test_crop_rejects_or_poisons_out_of_range_box_index():
    """a box whose image index is outside [0, N) must never read or scatter into foreign memory.  The
    drop-in wrapper validates a CPU-resident index like the reference's asserts (bilinear.py:122-123); with a device
    index the kernel fills that crop with NaN and drops its gradient, the other boxes are unaffected."""
    from agl import lib as L
    from models.bilinear import crop_bbox_batch
    feats = rn(2, 3, 16, 16)
    boxes = torch.tensor([[0.1, 0.1, 0.6, 0.7], [0.2, 0.0, 0.9, 0.5], [0.0, 0.3, 0.5, 1.0]])
    with pytest.raises(IndexError):
        crop_bbox_batch(dev(feats), dev(boxes), torch.tensor([0, 2, 1]), 8)
    with pytest.raises(IndexError):
        crop_bbox_batch(dev(feats), dev(boxes), torch.tensor([0, -1, 1]), 8)
    good = L.crop_fwd(dev(feats), dev(boxes), dev(torch.tensor([0, 1, 1])), 8, 8)
    bad = L.crop_fwd(dev(feats), dev(boxes), dev(torch.tensor([0, 7, 1])), 8, 8)
    torch.cuda.synchronize()
    assert torch.isnan(bad[1]).all() and torch.equal(bad[0], good[0]) and torch.equal(bad[2], good[2])
    gy = dev(rn(3, 3, 8, 8, seed=2))
    d_bad = L.crop_bwd(gy, dev(boxes), dev(torch.tensor([0, -3, 1])), (2, 3, 16, 16))
    gy2 = gy.clone()
    gy2[1] = 0
    d_ref = L.crop_bwd(gy2, dev(boxes), dev(torch.tensor([0, 1, 1])), (2, 3, 16, 16))
    close(d_bad, d_ref.cpu(), 1e-6, "gradient of the valid boxes only")


def test_loss_kernels_large_rows_and_bad_labels():
    """the attribute BCE takes any number of rows (was capped at 4096 by an LDS table); an out-of-range class
    label poisons the cross-entropy with NaN instead of reading out of bounds; the multi-block L1 equals torch."""
    from agl import losses as LS
    g = torch.Generator().manual_seed(5)
    rows, A = 5000, 106
    x = torch.randn(rows, A, generator=g)
    t = (torch.rand(rows, A, generator=g) < 0.02).float()
    t[::3] = 0                                          # un-annotated rows
    pw = torch.rand(A, generator=g) * 20 + 1
    sel = t.sum(1) != 0
    xr = x.clone().requires_grad_(True)
    ref = TF.binary_cross_entropy_with_logits(xr[sel], t[sel], pos_weight=pw)
    ref.backward()
    slot = torch.zeros(1, device=DEV)
    dx = LS.bce_posw(dev(x), dev(t), dev(pw), 1.0, slot)
    assert abs(float(slot) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    close(dx, xr.grad, 1e-5, "bce_posw gradient, 5000 rows")
    lg = torch.randn(6, 11, generator=g)
    lab = torch.tensor([0, 10, 3, 11, 2, -1])
    dl = LS.cross_entropy(dev(lg), dev(lab), 1.0, slot)
    assert torch.isnan(slot).all() and torch.isnan(dl[3]).all() and torch.isnan(dl[5]).all() and torch.isfinite(dl[0]).all()
    a, b = torch.randn(7, 3, 64, 64, generator=g), torch.randn(7, 3, 64, 64, generator=g)
    keep = torch.tensor([0., 0., 1., 1., 1., 1., 1.])
    ar = a.clone().requires_grad_(True)
    ref = (keep.view(-1, 1, 1, 1) * (ar - b).abs()).sum() / (3 * 64 * 64 * 5.0)
    ref.backward()
    da = LS.l1_rows(dev(a), dev(b), dev(keep), 1.0, 5.0, slot)
    assert abs(float(slot) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    close(da, ar.grad, 1e-6, "l1_rows gradient")


NORM_FOLD_CASES = [  # (mode: 0 BN / 1 affine BN / 2 ConditionalBN), N, Cin, H, Cout, ks, stride, pad, relu
    (2, 6, 64, 16, 128, 4, 2, 1, True), (2, 9, 32, 32, 64, 4, 2, 1, True), (1, 5, 32, 16, 64, 3, 1, 1, True), (0, 4, 48, 16, 96, 4, 2, 1, False),
    (2, 3, 64, 33, 128, 4, 2, 1, True), (2, 7, 32, 8, 64, 5, 1, 2, False), (1, 17, 64, 8, 128, 4, 2, 1, True), (2, 2, 128, 64, 256, 4, 2, 1, True),
    (2, 37, 64, 4, 128, 4, 2, 1, True)]      # (last: 4x4 -> 2x2, the crop encoder's last layer — sixteen whole 2x2 maps per weight-gradient tile)


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", NORM_FOLD_CASES)
def test_norm_folded_into_the_consuming_convolution(case, mode):
    """BASELINE north_star / SURVEY a2: conv(relu(CondBN(x))) with the normalise-modulate applied while the convolution stages its
    input (agl_norm_fold_table + agl_conv2d_fwd_fold + agl_conv2d_bwd_weight_fold + agl_norm_bwd_fold; F.norm_conv2d).  Against
    torch on the CPU (BatchNorm in training mode with the module's own affine / class-table parameters, then ReLU, then conv2d):
    output, input gradient, parameter gradients of the norm, weight gradient, running statistics; against the two-pass HIP form
    (AGL_NORM_FOLD off: same kernels' arithmetic up to the order of one multiplication); and that the folded call left the
    BatchNorm partial rows of ITS output for the next norm.  The transform is compiled into the 4x4 / stride-2 instantiations (every
    normalise -> convolution pair of the crop, layout and global encoders): maps of 64 down to 8 pixels, an odd size (33), all three
    norm kinds; the 3x3 / 5x5 stride-1 cases take F.norm_conv2d's two-pass fallback and must give the same numbers."""
    from agl import functional as F
    from agl import lib as L
    from agl import nn as A
    from agl.generator import ConditionalBatchNorm2d
    kind, N, Cin, H, Cout, ks, stride, pad, relu = case
    tol_y, tol_g = (2e-5, 2e-4) if mode == "split3" else (2e-2, 4e-2)
    torch.manual_seed(3)
    if kind == 2:
        norm = ConditionalBatchNorm2d(Cin, 7)
        norm.embed.weight.data[:, Cin:] = 0.3 * torch.randn(7, Cin)
    else:
        norm = A.BatchNorm2d(Cin, affine=kind == 1)
        if kind == 1:
            norm.weight.data = 1.0 + 0.3 * torch.randn(Cin)
            norm.bias.data = 0.2 * torch.randn(Cin)
    conv = A.Conv2d(Cin, Cout, kernel_size=ks, stride=stride, padding=pad, bias=kind != 2)
    labels = torch.randint(0, 7, (N,))
    x = rn(N, Cin, H, H) * 1.7 + 0.4
    OH = (H + 2 * pad - ks) // stride + 1
    gy = rn(N, Cout, OH, OH, seed=5)
    # torch reference on the CPU
    xr = x.clone().requires_grad_(True)
    bnr = torch.nn.BatchNorm2d(Cin, affine=False)
    xh = bnr(xr)
    if kind == 2:
        tab = norm.embed.weight.detach().clone().requires_grad_(True)
        gb = tab[labels]
        h = gb[:, :Cin].view(N, Cin, 1, 1) * xh + gb[:, Cin:].view(N, Cin, 1, 1)
        pr = [tab]
    elif kind == 1:
        gw, gbias = norm.weight.detach().clone().requires_grad_(True), norm.bias.detach().clone().requires_grad_(True)
        h = xh * gw.view(1, -1, 1, 1) + gbias.view(1, -1, 1, 1)
        pr = [gw, gbias]
    else:
        h, pr = xh, []
    wr = conv.weight.detach().clone().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True) if conv.bias is not None else None
    yr = TF.conv2d(torch.relu(h) if relu else h, wr, br, stride, pad)
    yr.backward(gy)
    flags = (L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3) | L.CONV_ANY_GRID
    res = {}
    for folded in (True, False):
        nd, cd = __import__("copy").deepcopy(norm).to(DEV), __import__("copy").deepcopy(conv).to(DEV)
        xd = dev(x).requires_grad_(True)
        prev, F.NORM_FOLD = F.NORM_FOLD, folded
        prev_e, F.EMIT_STATS = F.EMIT_STATS, True
        try:
            with L.conv_flags(flags):
                if folded:      # (the transform is compiled into the 4x4 / stride-2 family: the other cases pin the two-pass fallback)
                    assert L.conv_fold_ok(N, Cin, H, H, Cout, ks, stride, pad) == (ks == 4 and stride == 2), "4x4 / stride 2 must fold"
                y = F.norm_conv2d(xd, nd, dev(labels) if kind == 2 else None, cd, relu=relu, training=True)
                stats = F._LAST_STATS
                y.backward(dev(gy))
        finally:
            F.NORM_FOLD, F.EMIT_STATS = prev, prev_e
        torch.cuda.synchronize()
        bn = getattr(nd, "bn", nd)
        params = [nd.embed.weight.grad] if kind == 2 else ([nd.weight.grad, nd.bias.grad] if kind == 1 else [])
        res[folded] = (y.detach(), xd.grad, params, cd.weight.grad, cd.bias.grad if cd.bias is not None else None, bn.running_mean.clone(),
                       bn.running_var.clone(), int(bn.num_batches_tracked), stats)
    for folded, tag in ((True, "folded"), (False, "two passes")):
        y, dx, params, dw, db, rm, rv, nbt, stats = res[folded]
        close(y, yr, tol_y, f"{tag}: y")
        close(dx, xr.grad, tol_g, f"{tag}: dx")
        for a, r in zip(params, pr):
            close(a, r.grad, tol_g, f"{tag}: norm parameter gradient")
        close(dw, wr.grad, tol_g, f"{tag}: dw")
        if db is not None:
            close(db, br.grad, tol_g, f"{tag}: db")
        close(rm, bnr.running_mean, 1e-5, f"{tag}: running mean")
        close(rv, bnr.running_var, 1e-5, f"{tag}: running var")
        assert nbt == 1
    close(res[True][0], res[False][0], 1e-5 if mode == "split3" else 1e-2, "folded vs two passes: y")
    if mode == "bf16":      # the tight form of that statement: the two differ by single bf16 roundings of a few staged elements (tools/fold_diag.py)
        d = (res[True][0] - res[False][0]).double()
        rel_rms = float(d.pow(2).mean().sqrt() / res[False][0].double().pow(2).mean().sqrt().clamp_min(1e-30))
        assert rel_rms <= 2e-4, ("folded vs two passes: rms distance of y relative to its rms", rel_rms)
    close(res[True][1], res[False][1], 1e-4 if mode == "split3" else 2e-2, "folded vs two passes: dx")
    # the folded call leaves the partial rows of its own output (the next norm's statistics without a read of y)
    y, stats = res[True][0], res[True][8]
    if stats is not None:
        m1, r1 = L.bn_stats_from_partials(stats[2], stats[3], Cout, y.numel() // Cout, 1e-5, 0.1)
        m2, r2 = L.bn_stats(y, 1e-5, 0.1)
        close(m1, m2, 2e-6, "partials of the folded call: mean"); close(r1, r2, 2e-5, "partials of the folded call: rstd")


def test_folded_norm_at_a_mean_a_hundred_standard_deviations_out():
    """The fold's tables are fp32 (include/agl.h: shift = beta - mean * scale formed in double, rounded once), so v = fma(x, scale, shift)
    cancels two numbers of size |mean|/std against each other where the two-pass form subtracts first.  At |mean|/std = 100 that is an
    error of about 100 * 2^-24 = 6e-6 of a standard deviation per staged element: held here, in the exact (split) arithmetic where nothing
    else hides it, against a float64 restatement of BatchNorm -> ReLU -> conv — the folded output within 1e-4 of max|y| and within 8x
    the two-pass form's own distance (which has the fp32 mean's rounding in it too)."""
    from agl import functional as F
    from agl import lib as L
    from agl import nn as A
    N, Cin, H, Cout = 6, 64, 16, 128
    torch.manual_seed(11)
    norm = A.BatchNorm2d(Cin, affine=True)
    norm.weight.data = 1.0 + 0.3 * torch.randn(Cin)
    norm.bias.data = 0.2 * torch.randn(Cin)
    conv = A.Conv2d(Cin, Cout, kernel_size=4, stride=2, padding=1, bias=True)
    x = rn(N, Cin, H, H) + 100.0 * (1.0 + 0.1 * torch.randn(1, Cin, 1, 1))
    xd64 = x.double()
    mean = xd64.mean((0, 2, 3), keepdim=True)
    var = xd64.var((0, 2, 3), unbiased=False, keepdim=True)
    h = (xd64 - mean) / (var + 1e-5).sqrt() * norm.weight.double().view(1, -1, 1, 1) + norm.bias.double().view(1, -1, 1, 1)
    ref = TF.conv2d(torch.relu(h), conv.weight.double(), conv.bias.double(), 2, 1)
    out = {}
    for folded in (True, False):
        nd, cd = __import__("copy").deepcopy(norm).to(DEV), __import__("copy").deepcopy(conv).to(DEV)
        prev, F.NORM_FOLD = F.NORM_FOLD, folded
        try:
            with L.conv_flags(L.CONV_SPLIT3 | L.CONV_ANY_GRID), torch.no_grad():
                out[folded] = F.norm_conv2d(dev(x), nd, None, cd, relu=True, training=True).double().cpu()
        finally:
            F.NORM_FOLD = prev
    scale = float(ref.abs().max())
    e_fold, e_two = float((out[True] - ref).abs().max()) / scale, float((out[False] - ref).abs().max()) / scale
    assert e_two <= 3e-5, ("two passes at mean/std = 100", e_two)
    assert e_fold <= 1e-4 and e_fold <= 8 * max(e_two, 5e-6), ("folded at mean/std = 100", e_fold, e_two)


@pytest.mark.parametrize("consumer", ["convT", "conv5", "conv3"])
def test_spade_output_stored_as_bf16_gives_identical_results(consumer):
    """in bf16 arithmetic the SPADE-modulated tensor feeds ONE layer (a ConvTranspose2d(4,2,1) or a convolution);
    as one graph node (F.spade_modulate_then) it is stored as bf16 between them.  Its readers — the consumer's forward (dy role of the
    phase kernel / x role of the patch kernel), the consumer's weight gradient, the ReLU mask of the modulation's backward — round it
    to bf16 when staging it anyway, so everything must be BIT-IDENTICAL to the fp32-stored form: output, input gradient, gradient
    of the segmentation map, every parameter gradient, running statistics."""
    import copy
    from agl import functional as F
    from agl import lib as L
    from agl import nn as A
    from agl.generator import SPADE
    torch.manual_seed(5)
    N, Cc, S = 6, 64, {"convT": 32, "conv5": 64, "conv3": 16}[consumer]      # (segmentation map 8x8: block-class grids with / without the folded gather, plain form)
    sp = SPADE(Cc, 64)
    layer = (A.ConvTranspose2d(Cc, 48, kernel_size=4, stride=2, padding=1, bias=False) if consumer == "convT" else
             A.Conv2d(Cc, 64, kernel_size=5, padding=2, bias=False) if consumer == "conv5" else A.Conv2d(Cc, 128, kernel_size=3, padding=1))
    x, seg = rn(N, Cc, S, S) * 1.3 + 0.2, rn(N, 64, 8, 8, seed=2)
    res = []
    for y16 in (True, False):
        spd, ld = copy.deepcopy(sp).to(DEV), copy.deepcopy(layer).to(DEV)
        xd, sd = dev(x).requires_grad_(True), dev(seg).requires_grad_(True)
        prev, F.SPADE_Y16 = F.SPADE_Y16, y16
        try:
            with L.conv_flags(L.CONV_BF16 | L.CONV_ANY_GRID):
                if y16:
                    kind = "convT" if consumer == "convT" else "conv"
                    co = ld.weight.shape[1] if kind == "convT" else ld.weight.shape[0]
                    assert L.norm_output_as_bf16(N, Cc, S, S, kind, co, ld.kernel_size[0], 1, ld.padding[0]), "case must take the bf16 form"
                out = spd(xd, sd, relu=True, then=ld)
                out.backward(dev(rn(*out.shape, seed=7)))
        finally:
            F.SPADE_Y16 = prev
        torch.cuda.synchronize()
        res.append([out.detach(), xd.grad, sd.grad, ld.weight.grad] + [q.grad for q in spd.parameters()] +
                   ([ld.bias.grad] if getattr(ld, "bias", None) is not None else []) + [spd.param_free_norm.running_mean, spd.param_free_norm.running_var])
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), (consumer, i, float((a - b).abs().max()))
    # and the pair equals torch within the bf16 bars
    xr = x.clone().requires_grad_(True)
    spc, lc = copy.deepcopy(sp), copy.deepcopy(layer)
    bn = torch.nn.BatchNorm2d(Cc, affine=False)
    segu = TF.interpolate(seg, size=(S, S), mode="nearest")
    actv = torch.relu(TF.conv2d(segu, spc.mlp_shared[0].weight, spc.mlp_shared[0].bias, padding=1))
    gam = TF.conv2d(actv, spc.mlp_gamma.weight, spc.mlp_gamma.bias, padding=1)
    bet = TF.conv2d(actv, spc.mlp_beta.weight, spc.mlp_beta.bias, padding=1)
    h = torch.relu(bn(xr) * (1 + gam) + bet)
    ref = TF.conv_transpose2d(h, lc.weight, None, 2, 1) if consumer == "convT" else TF.conv2d(h, lc.weight, lc.bias, 1, lc.padding[0])
    close(res[0][0], ref, 3e-2, "SPADE + consumer vs torch (bf16 arithmetic)")


@pytest.mark.parametrize("consumer", ["conv5", "conv7to3"])
def test_spade_applied_by_the_staging_pass_of_its_consumer(consumer):
    """BASELINE north_star "SPADE normalization fused with the following conv" (normalization.py:97,106 in front of
    generator_obj_att128.py:588-597's c6 / c7): with gamma|beta on a class grid (8x8 segmentation map, 128-wide activation: 40 x 40
    cells, ~10 % of the map) the modulate + ReLU is applied by the convolution's own staging pass (F._SpadeFoldConv:
    agl_spade_cells, agl_conv2d_fwd_spade, agl_conv2d_bwd_weight_spade, agl_norm_bwd_spade) and the modulated tensor is never written.
    Every reader evaluates the stand-alone apply's expression (csrc/spade.h), so everything must be BIT-IDENTICAL to the node that
    stores the tensor as bf16 (F._SpadeThenConv): output, input gradient, gradient of the segmentation map, every parameter gradient,
    running statistics — and no agl_norm_apply_fwd launch may be left in the folded node."""
    import copy
    from agl import functional as F
    from agl import lib as L
    from agl import nn as A
    from agl.generator import SPADE
    torch.manual_seed(5)
    N, Cc, S = 3, 128, 128
    sp = SPADE(Cc, 64)
    layer = (A.Conv2d(Cc, 128, kernel_size=5, padding=2, bias=False) if consumer == "conv5" else A.Conv2d(Cc, 3, kernel_size=7, padding=3, bias=True))
    x, seg = rn(N, Cc, S, S) * 1.3 + 0.2, rn(N, 64, 8, 8, seed=2)
    gy = rn(N, layer.out_channels, S, S, seed=7)
    res, applies = [], []
    for fold in (True, False):
        spd, ld = copy.deepcopy(sp).to(DEV), copy.deepcopy(layer).to(DEV)
        xd, sd = dev(x).requires_grad_(True), dev(seg).requires_grad_(True)
        prev, F.SPADE_FOLD = F.SPADE_FOLD, fold
        names = []
        orig_call = L.call

        def spy(name, *a):
            names.append(name)
            return orig_call(name, *a)
        L.call = spy
        try:
            with L.conv_flags(L.CONV_BF16 | L.CONV_ANY_GRID):
                if fold:
                    assert L.conv_spade_ok(N, Cc, S, S, layer.out_channels, layer.kernel_size[0], 1, layer.padding[0]), "case must take the folded form"
                out = spd(xd, sd, relu=True, then=ld)
                out.backward(dev(gy))
        finally:
            F.SPADE_FOLD = prev
            L.call = orig_call
        torch.cuda.synchronize()
        applies.append(sum(1 for n in names if n.startswith("agl_norm_apply_fwd")))
        res.append([out.detach(), xd.grad, sd.grad, ld.weight.grad] + [q.grad for q in spd.parameters()] +
                   ([ld.bias.grad] if getattr(ld, "bias", None) is not None else []) + [spd.param_free_norm.running_mean, spd.param_free_norm.running_var])
    assert applies == [0, 1], ("agl_norm_apply_fwd launches (folded, stored)", applies)
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), (consumer, i, float((a - b).abs().max()))


def test_modules_vs_reference_op_fixtures(golden_dir):
    """The ConditionalBatchNorm2d / SPADE / spectrally-normalised discriminator-block vectors of tests/golden/ops_small.npz
    (outputs, input and parameter gradients, running statistics after 1 and 3 calls, spectral-norm u/v after k forwards —
    all produced by the imported REFERENCE modules) on the HIP path, with the closed-form weights of oracle/fill.py."""
    from oracle.fill import fill_state
    from models.generator_obj_att import ConditionalBatchNorm2d
    from models.spade.networks.normalization import SPADE
    from models.discriminator import OptimizedBlock, ResidualBlock, add_sn
    o = np.load(os.path.join(golden_dir, "ops_small.npz"))
    t = lambda k: torch.from_numpy(o[k])

    def filled(m):
        m.load_state_dict(fill_state(m.state_dict()))
        return m.to(DEV)

    def sn_w(m):      # what the owning discriminator does once per forward call (agl.discriminator._Discriminator._weights)
        from agl import functional as F
        mods = [q for q in m.modules() if getattr(q, "has_sn", False)]
        ws = F.spectral_norm_weights([q.weight_orig for q in mods], [q.weight_u for q in mods], [q.weight_v for q in mods], m.training)
        return {id(q): w for q, w in zip(mods, ws)}

    cbn = filled(ConditionalBatchNorm2d(6, 5))
    x = dev(t("cbn_x")).requires_grad_(True)
    y = cbn(x, dev(t("cbn_labels")))
    y.backward(dev(t("cbn_gy")))
    close(y, t("cbn_y"), 2e-5, "condbn y")
    close(x.grad, t("cbn_dx"), 2e-4, "condbn dx")
    close(cbn.embed.weight.grad, t("cbn_dembed"), 2e-4, "condbn dembed")
    close(cbn.bn.running_mean, t("cbn_rm1"), 1e-5, "condbn running_mean after 1")
    close(cbn.bn.running_var, t("cbn_rv1"), 1e-5, "condbn running_var after 1")
    for _ in range(2):
        cbn(dev(t("cbn_x") * 1.5 + 0.3), dev(t("cbn_labels")))
    close(cbn.bn.running_mean, t("cbn_rm3"), 1e-5, "condbn running_mean after 3")
    close(cbn.bn.running_var, t("cbn_rv3"), 1e-5, "condbn running_var after 3")

    for S in (8, 16):
        sp = filled(SPADE(16, 64))
        k = f"spade{S}_"
        x, seg = dev(t(k + "x")).requires_grad_(True), dev(t(k + "seg")).requires_grad_(True)
        y = sp(x, seg)
        y.backward(dev(t(k + "gy")))
        close(y, t(k + "y"), 2e-5, k + "y")
        close(x.grad, t(k + "dx"), 2e-4, k + "dx")
        close(seg.grad, t(k + "dseg"), 2e-4, k + "dseg")
        close(sp.mlp_shared[0].weight.grad, t(k + "dWshared"), 2e-4, k + "dWshared")
        close(sp.mlp_gamma.weight.grad, t(k + "dWgamma"), 2e-4, k + "dWgamma")
        close(sp.mlp_beta.bias.grad, t(k + "dbbeta"), 2e-4, k + "dbbeta")
        close(sp.param_free_norm.running_var, t(k + "rv1"), 1e-5, k + "running_var")

    for tag, down in (("opt_down", True), ("opt_flat", False)):
        m = filled(add_sn(OptimizedBlock(3, 8, downsample=down)))
        x = dev(t(f"d{tag}_x")).requires_grad_(True)
        y = m(x, sn_w(m))
        y.backward(dev(t(f"d{tag}_gy")))
        close(y, t(f"d{tag}_y"), 5e-5, tag + " y")
        close(x.grad, t(f"d{tag}_dx"), 5e-4, tag + " dx")
        close(m.resi[2].weight_orig.grad, t(f"d{tag}_dW2"), 5e-4, tag + " dW2")
        close(m.sc.weight_orig.grad, t(f"d{tag}_dWsc"), 5e-4, tag + " dWsc")
        close(m.resi[0].weight_u, t(f"d{tag}_u0"), 1e-5, tag + " u after 1 forward")
    m = filled(add_sn(ResidualBlock(8, 16, downsample=True)))
    x = dev(t("dres_x")).requires_grad_(True)
    y = m(x * 1.0, sn_w(m))
    y.backward(dev(t("dres_gy")))
    close(y, t("dres_y"), 5e-5, "D res y (aliased shortcut)")
    close(x.grad, t("dres_dx"), 5e-4, "D res dx")
    close(m.resi[3].weight_orig.grad, t("dres_dW3"), 5e-4, "D res dW3")
    close(m.sc.weight_orig.grad, t("dres_dWsc"), 5e-4, "D res dWsc")
    for k in range(2, 8):
        with torch.no_grad():
            m(dev(t("dres_x")).clone(), sn_w(m))
        if k in (3, 7):
            close(m.resi[3].weight_u, t(f"dres_u3_after{k}"), 2e-5, f"SN u after {k} forwards")
            close(m.resi[3].weight_v, t(f"dres_v3_after{k}"), 2e-5, f"SN v after {k} forwards")


PCONV_CASES = [  # N, Cin, H, Cout, ks  (stride 1, "same" padding): the tile geometries of csrc/pconv.hip; 3x3, 5x5 and 1x1
    (4, 64, 32, 128, 3), (3, 32, 16, 64, 3), (2, 48, 64, 80, 3), (5, 64, 8, 128, 3), (7, 32, 8, 200, 3), (9, 64, 4, 128, 3),
    (17, 32, 4, 64, 3), (2, 32, 16, 128, 5), (6, 48, 8, 64, 5), (3, 16, 32, 48, 5), (9, 64, 24, 128, 3), (7, 32, 40, 64, 3),
    (3, 64, 32, 128, 1), (5, 32, 16, 64, 1), (7, 48, 8, 200, 1), (19, 64, 4, 256, 1), (2, 16, 24, 48, 1),
    (9, 512, 4, 768, 3)]      # (last: 384 channel blocks — the weight gradient is written straight to dw by a single split)


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", PCONV_CASES)
def test_pconv_bf16_matrix_core_patch_kernel(case, mode):
    """The LDS-patch kernel on the bf16 matrix cores (csrc/pconv.hip), forward and input-gradient forms with every fused
    epilogue.  'bf16' (AGL_CONV_BF16): equal to an fp32 convolution of the bf16-rounded operands to fp32 accuracy (2e-5).
    'split3' (AGL_CONV_SPLIT3): fp32 operands as three bf16 terms, six products — must meet the SAME tolerance as the exact
    fp32 MFMA path against torch fp32 (2e-5 forward, 1e-4 input gradient), and against an fp64 convolution its error must not
    exceed 2x the error of the exact fp32 MFMA kernel on the same problem (+2e-7): an fp32-accurate path, not a reduced-precision one."""
    from agl import lib as L
    N, Cin, H, Cout, ks = case
    p = ks // 2
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    if mode == "bf16":
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        flags, tol = L.CONV_BF16 | L.CONV_ANY_GRID, 2e-5
    else:
        r = lambda t: t
        flags, tol = L.CONV_SPLIT3 | L.CONV_ANY_GRID, 2e-5
    xr, wr = r(x), r(w)
    y_ref = TF.conv2d(torch.relu(xr) if mode == "split3" else r(torch.relu(x)), wr, b, padding=p)
    xd, wd, bd = dev(x), dev(w), dev(b)
    with L.conv_flags(flags):
        y = L.conv2d_fwd(xd, wd, bd, 1, p, in_relu=True)
        close(y, y_ref, tol, "forward (input ReLU, bias)")
        base = rn(*y_ref.shape, seed=5)
        y2 = L.conv2d_fwd(xd, wd, None, 1, p, out=dev(base).clone(), accumulate=True)
        close(y2, base + TF.conv2d(xr, wr, None, padding=p), tol, "accumulate")
        y3 = L.conv2d_fwd(xd, wd, bd, 1, p, relu=True)
        close(y3, torch.relu(TF.conv2d(xr, wr, b, padding=p)), tol, "output ReLU")
        # input gradient = same kernel with flipped taps / swapped channel roles, masked by the consumer's ReLU
        gy = rn(*y_ref.shape, seed=7)
        gyr = r(gy)
        xg = xr.clone().requires_grad_(True)
        TF.conv2d(xg, wr, None, padding=p).backward(gyr)
        mask = rn(N, Cin, H, H, seed=9)
        dx = L.conv2d_bwd_data(dev(gy), wd, (H, H), 1, p, pos_mask=dev(mask))
        close(dx, xg.grad * (mask > 0), 1e-4 if mode == "split3" else 5e-5, "input gradient with ReLU mask")
        # weight gradient (16x16x32 MFMA, transposed LDS reads; csrc/pconv.hip pbww_k), plain / with input ReLU / accumulating
        wg = wr.clone().requires_grad_(True)
        TF.conv2d(xr, wg, None, padding=p).backward(gyr)
        dw = L.conv2d_bwd_weight(dev(gy), xd, ks, 1, p)
        close(dw, wg.grad, 1e-4, "weight gradient")
        # ... with the bias gradient formed by the same kernel from the dy tiles it stages (fresh, then accumulating)
        db = torch.full((Cout,), float("nan"), device=DEV)
        dw_b = L.conv2d_bwd_weight(dev(gy), xd, ks, 1, p, dbias=db)
        assert torch.equal(dw_b, dw)
        close(db, gy.double().sum((0, 2, 3)).float(), 2e-5, "bias gradient from the weight-gradient kernel")
        db2 = dev(b).clone()
        L.conv2d_bwd_weight(dev(gy), xd, ks, 1, p, out=dev(base_w0 := rn(Cout, Cin, ks, ks, seed=12)).clone(), accumulate=True, dbias=db2)
        close(db2, b + gy.double().sum((0, 2, 3)).float(), 2e-5, "bias gradient, accumulating")
        wg2 = wr.clone().requires_grad_(True)
        TF.conv2d(torch.relu(xr) if mode == "split3" else r(torch.relu(x)), wg2, None, padding=p).backward(gyr)
        base_w = rn(Cout, Cin, ks, ks, seed=11)
        dw2 = L.conv2d_bwd_weight(dev(gy), xd, ks, 1, p, in_relu=True, out=dev(base_w).clone(), accumulate=True)
        close(dw2, base_w + wg2.grad, 1e-4, "weight gradient (input ReLU, accumulate)")
    if mode == "split3":
        # accuracy class: against an fp64 convolution the split path must be as accurate as the exact fp32 MFMA chain
        # (v_mfma_f32_32x32x2_f32, flags 0) on the same problem — it is not a reduced-precision mode
        y64 = TF.conv2d(torch.relu(x).double(), w.double(), b.double(), padding=p)
        y_exact = L.conv2d_fwd(xd, wd, bd, 1, p, in_relu=True)
        e_split = float((y.cpu().double() - y64).abs().max())
        e_exact = float((y_exact.cpu().double() - y64).abs().max())
        assert e_split <= 2.0 * e_exact + 2e-7 * float(y64.abs().max()), (e_split, e_exact)


@pytest.mark.parametrize("scales", [(1e-8, 1.0, 1.0), (1e4, 1.0, 1.0), (1.0, 1e-8, 1e4), (1e4, 1e4, 1e-8), (1e-30, 1e-5, 1.0), (1e25, 1e8, 1e-20)])
def test_split_products_accuracy_class_at_extreme_operand_magnitudes(scales):
    """The split arithmetic carries fp16 terms (5 exponent bits) under power-of-two block scales found when an operand is staged or
    packed, so its accuracy class must not depend on the operands' magnitude: activations, weights and output gradients scaled by
    1e-8 ... 1e4 (and far beyond: 1e-30, 1e25), plus one tensor whose channels span 2^40 in magnitude inside every staged block.  Same
    gate as the unit-magnitude tests: distance to fp64 <= 2x that of the exact fp32 MFMA kernel (+2e-7 of the result's largest value),
    forward, both input-gradient forms (stride 1, 4x4 / stride 2 phases) and the weight gradient, short and long reductions."""
    from agl import lib as L
    xs, ws, gs = scales
    g = torch.Generator().manual_seed(21)
    for (N, Cin, H, Cout, ks, st, pad) in ((9, 64, 16, 128, 3, 1, 1), (6, 512, 8, 256, 3, 1, 1), (8, 256, 16, 512, 4, 2, 1), (12, 128, 8, 256, 5, 1, 2)):
        x = torch.randn(N, Cin, H, H, generator=g) * xs
        w = torch.randn(Cout, Cin, ks, ks, generator=g) * (ws / (Cin * ks * ks) ** 0.5)
        OH = (H + 2 * pad - ks) // st + 1
        dy = torch.randn(N, Cout, OH, OH, generator=g) * gs
        x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
        y64 = TF.conv2d(x64, w64, None, stride=st, padding=pad)
        y64.backward(dy.double())
        refs = (y64.detach(), x64.grad, w64.grad)
        xd, wd, dyd = dev(x), dev(w), dev(dy)
        out = {}
        for name, flags in (("exact", 0), ("split", L.CONV_SPLIT3 | L.CONV_ANY_GRID)):
            with L.conv_flags(flags):
                out[name] = (L.conv2d_fwd(xd, wd, None, st, pad), L.conv2d_bwd_data(dyd, wd, (H, H), st, pad), L.conv2d_bwd_weight(dyd, xd, ks, st, pad))
                if name == "split":
                    assert L.load().agl_conv2d_last_pipe() == 3, "the call must run on the split matrix-core kernels"
        for i, what in enumerate(("forward", "input gradient", "weight gradient")):
            e_exact = float((out["exact"][i].cpu().double() - refs[i]).abs().max())
            e_split = float((out["split"][i].cpu().double() - refs[i]).abs().max())
            assert e_split <= 2.0 * e_exact + 2e-7 * float(refs[i].abs().max()), (scales, (N, Cin, H, Cout, ks, st), what, e_split, e_exact)
    # a wide range INSIDE every staged block: channel c of x scaled by 2^(-(c % 41)); every 16-channel chunk then spans > 2^15
    N, Cin, H, Cout = 6, 128, 16, 64
    chs = torch.pow(2.0, -(torch.arange(Cin) % 41).float()).view(1, Cin, 1, 1)
    x = torch.randn(N, Cin, H, H, generator=g) * chs
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (Cin * 9) ** 0.5)
    y64 = TF.conv2d(x.double(), w.double(), None, padding=1)
    with L.conv_flags(0):
        e_exact = float((L.conv2d_fwd(dev(x), dev(w), None, 1, 1).cpu().double() - y64).abs().max())
    with L.conv_flags(L.CONV_SPLIT3 | L.CONV_ANY_GRID):
        e_split = float((L.conv2d_fwd(dev(x), dev(w), None, 1, 1).cpu().double() - y64).abs().max())
    assert e_split <= 2.0 * e_exact + 2e-7 * float(y64.abs().max()), ("channel range 2^40", e_split, e_exact)


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", [(5, 32, 64, 64, 4, 2, 1), (9, 64, 16, 128, 4, 2, 1), (17, 32, 8, 64, 4, 2, 1), (3, 16, 24, 48, 5, 1, 2),
                                  (37, 32, 4, 80, 4, 2, 1), (4, 64, 32, 128, 4, 2, 1), (3, 32, 33, 64, 4, 2, 1), (2, 48, 64, 200, 5, 1, 2),
                                  (6, 32, 16, 64, 3, 1, 1)])
def test_conv_emits_batchnorm_partials(case, mode):
    """agl_conv2d_fwd_stats: the matrix-core convolution leaves per-channel (count, mean, M2) rows of the output it stores;
    agl_bn_stats_from_partials must then agree with agl_bn_stats run on that output (mean, rstd, running statistics, counter),
    and the output itself must equal the plain forward.  The rows are compiled into the instantiations whose layers are followed by
    a BatchNorm on the path — the 4x4 / stride-2 family (every tile geometry: 32 / 8 / 4 / 2-pixel output maps, an odd input) and the
    bf16 5x5 forms; elsewhere (the 3x3 case, 5x5 in split mode) the call reports no rows and the caller uses agl_bn_stats."""
    from agl import lib as L
    N, Cin, H, Cout, ks, stride, pad = case
    has_rows = (ks == 4 and stride == 2) or (ks == 5 and mode == "bf16")
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2) * 3.0
    flags = (L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3) | L.CONV_ANY_GRID
    xd, wd, bd = dev(x), dev(w), dev(b)
    with L.conv_flags(flags):
        y0 = L.conv2d_fwd(xd, wd, bd, stride, pad, in_relu=True)
        y, part, rows = L.conv2d_fwd_stats(xd, wd, bd, stride, pad, in_relu=True)
    assert torch.equal(y, y0)
    if not has_rows:
        assert part is None and rows == 0
        return
    assert part is not None and rows > 0, "the matrix-core kernel did not take this shape"
    C_, cnt = y.shape[1], y.numel() // y.shape[1]
    rm = [dev(rn(C_, seed=4)).clone() for _ in range(2)]
    rv = [dev(rn(C_, seed=5).abs() + 0.5).clone() for _ in range(2)]
    nb = [torch.tensor(7, device=DEV, dtype=torch.int64) for _ in range(2)]
    m1, r1 = L.bn_stats_from_partials(part, rows, C_, cnt, 1e-5, 0.1, rm[0], rv[0], nb[0])
    m2, r2 = L.bn_stats(y, 1e-5, 0.1, rm[1], rv[1], nb[1])
    close(m1, m2, 2e-6, "mean"); close(r1, r2, 2e-5, "rstd")
    close(rm[0], rm[1], 2e-6, "running mean"); close(rv[0], rv[1], 2e-5, "running var")
    assert int(nb[0]) == int(nb[1]) == 8
    with L.conv_flags(0):           # exact-fp32 kernels do not produce the rows: the caller falls back to agl_bn_stats
        _, part0, rows0 = L.conv2d_fwd_stats(xd, wd, bd, stride, pad)
    assert part0 is None and rows0 == 0


@pytest.mark.parametrize("case", [(4, 64, 32, 128, 3), (3, 32, 16, 64, 3), (5, 64, 8, 128, 3), (2, 48, 64, 80, 3), (2, 32, 16, 128, 5),
                                  (6, 48, 8, 64, 5), (9, 64, 24, 128, 3)])
def test_pconv_eight_wave_workgroups_equal_four_wave(case):
    """AGL_CONV_W8 (taken automatically on grids below 512 workgroups, as in these cases, and selectable per call): the
    split-mode stride-1 3x3 / 5x5 kernels with 512-thread workgroups — same tile, same LDS image, same order of accumulation
    per output element — must reproduce the 256-thread form bit for bit (forward with the statistics rows, and the input
    gradient), and the statistics rows must finalise to the same mean / rstd."""
    from agl import lib as L
    N, Cin, H, Cout, ks = case
    p = ks // 2
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    gy = rn(N, Cout, H, H, seed=3)
    outs = []
    for big_grid_form in (False, True):
        # the 4-wave form needs a grid of >= 512 workgroups to be chosen: force it through the reference path of a larger batch?  No —
        # compare against torch instead for the 4-wave form's tolerance, and require 8-wave == explicit W8 flag on the same small grid.
        flags = L.CONV_SPLIT3 | L.CONV_ANY_GRID | (L.CONV_W8 if big_grid_form else 0)
        with L.conv_flags(flags):
            y, part, rows = L.conv2d_fwd_stats(dev(x), dev(w), dev(b), 1, p, in_relu=True)
            dx = L.conv2d_bwd_data(dev(gy), dev(w), (H, H), 1, p)
            m, r = L.bn_stats_from_partials(part, rows, Cout, y.numel() // Cout, 1e-5, 0.1) if part is not None else (None, None)
        outs.append((y, dx, m, r))
    (y0, dx0, m0, r0), (y1, dx1, m1, r1) = outs
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    close(y0, TF.conv2d(torch.relu(x), w, b, padding=p), 2e-5, "eight-wave forward vs torch")
    if m0 is not None and m1 is not None:
        close(m0, m1, 1e-6, "mean from the statistics rows")
        close(r0, r1, 1e-5, "rstd from the statistics rows")


def test_split_products_accuracy_class_of_gradients_at_config2_extents():
    """AGL_CONV_SPLIT3 is used by the weight-gradient kernel (pbww_k: reductions over ~4e5 pixels) and by the phase-mode
    stride-2 input gradient as well as by the forward kernel; the accuracy-class check of the forward (error vs an fp64
    result <= 2x the error of the exact fp32 MFMA kernel on the same problem) is repeated here for both, at BASELINE config 2
    extents: N = 393 objects, 32x32 maps, 64 -> 128 channels (3x3 stride 1 weight gradient; 4x4 stride 2 input gradient and
    weight gradient)."""
    from agl import lib as L
    N, H = 393, 32
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, 64, H, H, generator=g)
    for ks, stride, pad, Cout in ((3, 1, 1, 128), (4, 2, 1, 128)):
        OH = (H + 2 * pad - ks) // stride + 1
        w = torch.randn(Cout, 64, ks, ks, generator=g) * (1.0 / (64 * ks * ks) ** 0.5)
        dy = torch.randn(N, Cout, OH, OH, generator=g)
        xd, wd, dyd = dev(x), dev(w), dev(dy)
        x64 = x.double().requires_grad_(True)
        w64 = w.double().requires_grad_(True)
        TF.conv2d(x64, w64, None, stride=stride, padding=pad).backward(dy.double())
        res = {}
        for name, flags in (("exact", 0), ("split", L.CONV_SPLIT3)):
            with L.conv_flags(flags):
                res[name] = (L.conv2d_bwd_weight(dyd, xd, ks, stride, pad).cpu().double(),
                             L.conv2d_bwd_data(dyd, wd, (H, H), stride, pad).cpu().double())
        for i, (what, ref) in enumerate((("weight gradient", w64.grad), ("input gradient", x64.grad))):
            e_exact = float((res["exact"][i] - ref).abs().max())
            e_split = float((res["split"][i] - ref).abs().max())
            assert e_split <= 2.0 * e_exact + 2e-7 * float(ref.abs().max()), (ks, stride, what, e_split, e_exact)


@pytest.mark.parametrize("mode", ["bf16", "split3"])
def test_packed_weight_cache_and_divisor(mode):
    """agl_conv2d_pack_weights + the packed_w / packed_div arguments (agl.lib.WeightSrc): a convolution that reads pre-packed
    weights must equal the one that packs per call — forward 3x3 / 1x1 / 4x4 stride 2, the "same" input gradient, the stride-2
    input gradient (ConvTranspose forward) — bit for bit without a divisor; with a divisor (spectral norm: packed weight_orig,
    device scalar sigma) equal to the convolution with w / sigma to fp32 rounding; the cache must re-pack when the weight version
    changes and only then."""
    from agl import lib as L
    flags = (L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3) | L.CONV_ANY_GRID
    ver = [0]
    # (the 33 x 33 case: odd-sized input of a 4x4 / stride-2 layer — its last row and column come from phase_edge_k, which reads the
    #  unpacked, already divided weights: ADVICE r3, the divisor must not be applied twice there)
    for (N, Cin, H, Cout, ks, stride, pad) in ((6, 64, 16, 64, 3, 1, 1), (5, 64, 8, 128, 1, 1, 0), (4, 64, 16, 64, 4, 2, 1), (3, 48, 32, 80, 5, 1, 2),
                                               (3, 64, 33, 64, 4, 2, 1)):
        x, w = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5)
        OH = (H + 2 * pad - ks) // stride + 1
        dy = rn(N, Cout, OH, OH, seed=3)
        owner = torch.nn.Parameter(dev(w).clone())
        src = L.WeightSrc(owner, lambda: ver[0])
        sigma = dev(torch.tensor([1.7]))
        src_sn = L.WeightSrc(owner, lambda: ver[0], base=owner, div=sigma, tag="sn")
        xd, dyd = dev(x), dev(dy)
        with L.conv_flags(flags):
            p0 = L.PACK_STATS["packs"]
            y0 = L.conv2d_fwd(xd, owner.data, None, stride, pad)
            y1 = L.conv2d_fwd(xd, owner.data, None, stride, pad, wsrc=src)
            y2 = L.conv2d_fwd(xd, owner.data, None, stride, pad, wsrc=src)
            assert torch.equal(y0, y1) and torch.equal(y0, y2)
            assert L.PACK_STATS["packs"] == p0 + 1, "second call must hit the cache"
            dx0 = L.conv2d_bwd_data(dyd, owner.data, (H, H), stride, pad)
            dx1 = L.conv2d_bwd_data(dyd, owner.data, (H, H), stride, pad, wsrc=src)
            assert torch.equal(dx0, dx1)
            # divisor: w_sn = w / sigma is what the caller hands over as `w`; the kernels read the packed w and divide
            w_sn = owner.data / sigma
            ys = L.conv2d_fwd(xd, w_sn, None, stride, pad, wsrc=src_sn)
            close(ys, L.conv2d_fwd(xd, w_sn, None, stride, pad), 2e-6 if mode == "split3" else 1e-2, "forward with divisor")
            close(ys, y0 / 1.7, 1e-6, "forward with divisor == plain / sigma")
            assert L.bwd_data_packed_bytes(N, Cin, H, H, Cout, OH, OH, ks, stride, pad) > 0, "case must exercise the packed input-gradient path"
            dxs = L.conv2d_bwd_data(dyd, w_sn, (H, H), stride, pad, wsrc=src_sn)
            # (odd-sized input in bf16 arithmetic: the edge kernel rounds w / sigma to bf16, the phases round w and divide afterwards)
            close(dxs, dx0 / 1.7, 1e-2 if (H % 2 == 1 and mode == "bf16") else 1e-6, "input gradient with divisor == plain / sigma")
            if H % 2 == 1:      # last row / column alone (1 / (2W) of the elements: an error there must not hide in the max)
                close(dxs[:, :, -1, :], dx0[:, :, -1, :] / 1.7, 1e-2 if mode == "bf16" else 2e-6, "edge row with divisor")
                close(dxs[:, :, :, -1], dx0[:, :, :, -1] / 1.7, 1e-2 if mode == "bf16" else 2e-6, "edge column with divisor")
            # new weight version: the cache must follow
            owner.data.mul_(-0.5)
            ver[0] += 1
            y3 = L.conv2d_fwd(xd, owner.data, None, stride, pad, wsrc=src)
            assert torch.equal(y3, L.conv2d_fwd(xd, owner.data, None, stride, pad))
            close(y3, -0.5 * y0, 1e-5, "re-packed after the version change")


FEW_BWW_CASES = [  # N, Cin, H, Cout, ks, pad, in_relu: weight gradients with <= 4 channels on one side (csrc/few.hip)
    (5, 3, 32, 64, 3, 1, 0), (4, 3, 32, 64, 1, 0, 0), (3, 3, 32, 64, 7, 3, 0), (2, 3, 64, 64, 3, 1, 1), (2, 3, 64, 80, 1, 0, 0),
    (2, 4, 32, 32, 3, 1, 0), (2, 1, 32, 48, 5, 2, 1), (1, 3, 128, 64, 5, 2, 0), (3, 2, 64, 128, 7, 3, 0), (2, 3, 32, 16, 3, 0, 0),
    (2, 64, 64, 3, 7, 3, 0), (3, 32, 32, 3, 3, 1, 0), (2, 48, 64, 1, 5, 2, 0)]   # (last three: few OUTPUT channels, roles swapped)


@pytest.mark.parametrize("case", FEW_BWW_CASES)
def test_few_channel_weight_gradient(case):
    """RGB-side layers (OptimizedBlock 3->64 discriminator.py:29-60, CropEncoder c1 generator_obj_att.py:367, decoder c4 64->3):
    the weight gradient as a 64 x (Cin*ks^2) MFMA product streamed over the pixels, exact fp32; plain and accumulating."""
    from agl import lib as L
    N, Cin, H, Cout, ks, pad, in_relu = case
    x, w = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1)
    xin = torch.relu(x) if in_relu else x
    OH = H + 2 * pad - ks + 1
    gy = rn(N, Cout, OH, OH, seed=3)
    wg = w.clone().requires_grad_(True)
    TF.conv2d(xin, wg, None, padding=pad).backward(gy)
    for flags in (0, L.CONV_SPLIT3):
        with L.conv_flags(flags):
            dw = L.conv2d_bwd_weight(dev(gy), dev(x), ks, 1, pad, in_relu=bool(in_relu))
            close(dw, wg.grad, 2e-5, f"weight gradient flags={flags}")
            base = rn(Cout, Cin, ks, ks, seed=5)
            dw2 = L.conv2d_bwd_weight(dev(gy), dev(x), ks, 1, pad, in_relu=bool(in_relu), out=dev(base).clone(), accumulate=True)
            close(dw2, base + wg.grad, 2e-5, f"accumulating flags={flags}")


@pytest.mark.parametrize("mode", ["bf16", "split3"])
def test_pconv_reduction_split(mode):
    """Grids that cannot fill the chip but have a long reduction (input gradients of the ConvLSTM recurrence steps: few images,
    512 -> 128 channels) run the matrix-core kernel with the channel chunks cut over blockIdx.z; the raw partial outputs are
    summed, and bias / ReLU mask / accumulate / ReLU applied, by the split-K epilogue.  No AGL_CONV_ANY_GRID here: the plan
    itself must choose the split."""
    from agl import lib as L
    N, Cin, H, Cout, ks = 48, 256, 8, 128, 3
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mode == "bf16" else (lambda t: t)
    flags = L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3
    xr, wr = r(x), r(w)
    xd, wd, bd = dev(x), dev(w), dev(b)
    with L.conv_flags(flags):
        y = L.conv2d_fwd(xd, wd, bd, 1, 1, relu=True)
        close(y, torch.relu(TF.conv2d(xr, wr, b, padding=1)), 2e-5, "forward, bias + ReLU through the slab reduction")
        base = rn(N, Cout, H, H, seed=5)
        y2 = L.conv2d_fwd(xd, wd, None, 1, 1, out=dev(base).clone(), accumulate=True)
        close(y2, base + TF.conv2d(xr, wr, None, padding=1), 2e-5, "accumulate")
        gy = rn(N, Cout, H, H, seed=7)
        xg = xr.clone().requires_grad_(True)
        TF.conv2d(xg, wr, None, padding=1).backward(r(gy))
        mask = rn(N, Cin, H, H, seed=9)
        dx = L.conv2d_bwd_data(dev(gy), wd, (H, H), 1, 1, pos_mask=dev(mask))
        close(dx, xg.grad * (mask > 0), 1e-4 if mode == "split3" else 5e-5, "input gradient with ReLU mask")


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", [(32, 256, 8, 128), (24, 512, 16, 64)])
def test_pconv_phase_reduction_split(case, mode):
    """The paired-phase stride-2 input gradient on a grid that cannot fill the chip: the plan cuts the reduction channels over
    blockIdx.z (no AGL_CONV_ANY_GRID), the slab reduction applies mask / accumulate."""
    from agl import lib as L
    N, Cout, OH, Cin = case            # dy: (N, Cout, OH, OH); dx: (N, Cin, 2*OH, 2*OH)
    w = rn(Cout, Cin, 4, 4, seed=1) * (1.0 / (Cout * 4) ** 0.5)
    gy = rn(N, Cout, OH, OH, seed=3)
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mode == "bf16" else (lambda t: t)
    ref = TF.conv_transpose2d(r(gy), r(w), None, stride=2, padding=1)
    mask, base = rn(N, Cin, 2 * OH, 2 * OH, seed=5), rn(N, Cin, 2 * OH, 2 * OH, seed=6)
    with L.conv_flags(L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3):
        dx = L.conv2d_bwd_data(dev(gy), dev(w), (2 * OH, 2 * OH), 2, 1)
        dx2 = L.conv2d_bwd_data(dev(gy), dev(w), (2 * OH, 2 * OH), 2, 1, pos_mask=dev(mask), out=dev(base).clone(), accumulate=True)
    close(dx, ref, 5e-5 if mode == "bf16" else 2e-5, "stride-2 input gradient (reduction split)")
    close(dx2, base + ref * (mask > 0), 5e-5, "with ReLU mask and accumulation")


def test_pconv_upsampled_input():
    """Nearest up-sampling folded into the patch staging (SPADE mlp_shared reads the 8x8 map up-sampled, normalization.py:100)."""
    from agl import lib as L
    x, w = rn(3, 64, 8, 8), rn(128, 64, 3, 3, seed=1) * 0.05
    for flags, r in ((L.CONV_BF16 | L.CONV_ANY_GRID, lambda t: t.to(torch.bfloat16).to(torch.float32)), (L.CONV_SPLIT3 | L.CONV_ANY_GRID, lambda t: t)):
        for up in (1, 2):
            ref = TF.conv2d(TF.interpolate(r(x), scale_factor=2 ** up, mode="nearest"), r(w), None, padding=1)
            with L.conv_flags(flags):
                y = L.conv2d_fwd(dev(x), dev(w), None, 1, 1, up=up)
            close(y, ref, 2e-5, f"up={up} flags={flags}")


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", [(4, 64, 32, 128, 4, 1), (5, 32, 16, 64, 4, 1), (9, 16, 8, 96, 4, 1), (4, 64, 33, 128, 3, 0), (6, 32, 17, 64, 3, 0),
                                  (10, 32, 9, 96, 3, 0), (2, 128, 64, 64, 4, 1), (3, 48, 66, 64, 4, 1), (37, 64, 5, 128, 3, 0), (21, 32, 4, 96, 4, 1)])
def test_pconv_stride2_forward(case, mode):
    """Stride-2 forward forms of the bf16-matrix-core patch kernel (4x4/pad 1: layout- and crop-encoder layers; 3x3/pad 0: the
    box form of the down-sampling discriminator blocks): the patch is split by column parity in LDS.  Same tolerances as the
    stride-1 cases, with bias / input ReLU / output ReLU."""
    from agl import lib as L
    N, Cin, H, Cout, ks, p = case
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, ks, ks, seed=1) * (1.0 / (Cin * ks * ks) ** 0.5), rn(Cout, seed=2)
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mode == "bf16" else (lambda t: t)
    yr = TF.relu(TF.conv2d(r(TF.relu(x)), r(w), b, stride=2, padding=p))
    with L.conv_flags((L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3) | L.CONV_ANY_GRID):   # (small tensors: below the occupancy threshold)
        y = L.conv2d_fwd(dev(x), dev(w), dev(b), 2, p, in_relu=True, relu=True)
        gy = rn(*yr.shape, seed=7)
        dw = L.conv2d_bwd_weight(dev(gy), dev(x), ks, 2, p)
    close(y, yr, 2e-5, "stride-2 forward")
    wg = r(w).clone().requires_grad_(True)
    TF.conv2d(r(x), wg, None, stride=2, padding=p).backward(r(gy))
    close(dw, wg.grad, 1e-4, "stride-2 weight gradient")


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", [(4, 64, 32, 128), (5, 96, 16, 64), (9, 128, 8, 80), (17, 64, 4, 256), (3, 48, 64, 64), (70, 128, 2, 256)])
def test_pconv_stride2_input_gradient_phases(case, mode):
    """4x4 / stride-2 / pad-1 input gradient (and ConvTranspose2d(4,2,1) forward) as four 2x2-tap phases on the bf16-matrix-core
    patch kernel, against torch: plain, and with the consumer's ReLU mask + accumulation."""
    from agl import lib as L
    N, Cout, OH, Cin = case            # dy: (N, Cout, OH, OH); dx: (N, Cin, 2*OH, 2*OH)
    w = rn(Cout, Cin, 4, 4, seed=1) * (1.0 / (Cout * 4) ** 0.5)
    gy = rn(N, Cout, OH, OH, seed=3)
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mode == "bf16" else (lambda t: t)
    ref = TF.conv_transpose2d(r(gy), r(w), None, stride=2, padding=1)
    mask, base = rn(N, Cin, 2 * OH, 2 * OH, seed=5), rn(N, Cin, 2 * OH, 2 * OH, seed=6)
    with L.conv_flags((L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3) | L.CONV_ANY_GRID):   # (small tensors: below the occupancy threshold)
        dx = L.conv2d_bwd_data(dev(gy), dev(w), (2 * OH, 2 * OH), 2, 1)
        dx2 = L.conv2d_bwd_data(dev(gy), dev(w), (2 * OH, 2 * OH), 2, 1, pos_mask=dev(mask), out=dev(base).clone(), accumulate=True)
    close(dx, ref, 5e-5 if mode == "bf16" else 2e-5, "stride-2 input gradient")
    close(dx2, base + ref * (mask > 0), 5e-5, "with ReLU mask and accumulation")


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", [(4, 64, 16, 128), (5, 96, 32, 64), (6, 32, 8, 80)])
def test_pconv_stride2_input_gradient_odd_sized_input(case, mode):
    """The layout encoder's c3 (generator_obj_att.py:479) maps a 33x33 (64 px) / 65x65 (128 px) input to 16x16 / 32x32: its input
    gradient has an ODD extent 2*OH+1.  Rows / columns 0..2*OH-1 come from the paired-phase matrix-core kernel (output pitch
    2*OH+1: under-aligned 16-byte stores), the last row and column from phase_edge_k — against torch's conv2d input gradient,
    plain and with ReLU mask + accumulation; the last row / column are checked on their own as well."""
    from agl import lib as L
    N, Cout, OH, Cin = case            # dy: (N, Cout, OH, OH); dx: (N, Cin, 2*OH+1, 2*OH+1)
    IH = 2 * OH + 1
    w = rn(Cout, Cin, 4, 4, seed=1) * (1.0 / (Cout * 4) ** 0.5)
    gy = rn(N, Cout, OH, OH, seed=3)
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mode == "bf16" else (lambda t: t)
    xg = torch.zeros(N, Cin, IH, IH, requires_grad=True)
    TF.conv2d(xg, r(w), None, stride=2, padding=1).backward(r(gy))
    ref = xg.grad
    assert float(ref[:, :, -1].abs().max()) > 0 and float(ref[:, :, :, -1].abs().max()) > 0
    mask, base = rn(N, Cin, IH, IH, seed=5), rn(N, Cin, IH, IH, seed=6)
    flags = (L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3) | L.CONV_ANY_GRID
    with L.conv_flags(flags):
        assert L.bwd_data_packed_bytes(N, Cin, IH, IH, Cout, OH, OH, 4, 2, 1) > 0, "the phase kernel must take the odd-sized form"
        dx = L.conv2d_bwd_data(dev(gy), dev(w), (IH, IH), 2, 1)
        dx2 = L.conv2d_bwd_data(dev(gy), dev(w), (IH, IH), 2, 1, pos_mask=dev(mask), out=dev(base).clone(), accumulate=True)
    tol = 5e-5 if mode == "bf16" else 2e-5
    close(dx, ref, tol, "odd-sized stride-2 input gradient")
    close(dx[:, :, -1], ref[:, :, -1], tol, "last row")
    close(dx[:, :, :, -1], ref[:, :, :, -1], tol, "last column")
    close(dx2, base + ref * (mask > 0), 5e-5, "with ReLU mask and accumulation")


@pytest.mark.parametrize("case", [(3, 64, 32, 64, 64), (2, 32, 16, 128, 128), (5, 128, 64, 32, 64), (2, 64, 64, 64, 72)])
@pytest.mark.parametrize("in_relu,relu,bias", [(False, False, False), (True, True, True)])
def test_channel_blocked_bf16_convolution_prototype(case, in_relu, relu, bias):
    """AGL_CONV_BLOCKED (include/agl.h, prototype): a 3x3 "same" convolution whose bf16 x and y are channel-blocked [N][C/8][H][W][8].
    The kernel stages and stores 16-byte pieces of 8 channels; products, accumulation order and the final rounding are those of the
    NCHW bf16-in / bf16-out call, so the two must agree bit for bit — and both with torch on the bf16-rounded operands."""
    from agl import lib as L
    N, Cin, H, W, Cout = case
    x, w, b = rn(N, Cin, H, W), rn(Cout, Cin, 3, 3, seed=1) * (1.0 / (Cin * 9) ** 0.5), (rn(Cout, seed=2) if bias else None)
    r = lambda t: t.to(torch.bfloat16).to(torch.float32)
    ref = TF.conv2d(torch.relu(r(x)) if in_relu else r(x), r(w), b, padding=1)
    if relu:
        ref = torch.relu(ref)
    with L.conv_flags(L.CONV_BF16 | L.CONV_ANY_GRID):
        xb = L.to_blocked(dev(x))
        assert torch.equal(L.from_blocked(xb), dev(x).to(torch.bfloat16))
        assert L.load().agl_conv2d_fwd_takes_blocked(N, Cin, H, W, Cout, 3, 1, 1, L.CONV_FLAGS | L.CONV_X_BF16 | L.CONV_Y_BF16 | L.CONV_BLOCKED)
        yb = L.conv2d_fwd_blocked(xb, dev(w), dev(b) if bias else None, in_relu=in_relu, relu=relu)
        y16 = L.conv2d_fwd(dev(x).to(torch.bfloat16), dev(w), dev(b) if bias else None, 1, 1, in_relu=in_relu, relu=relu, out_bf16=True)
    assert tuple(yb.shape) == (N, Cout // 8, H, W, 8) and yb.dtype == torch.bfloat16
    assert torch.equal(L.from_blocked(yb), y16), float((L.from_blocked(yb).float() - y16.float()).abs().max())
    close(L.from_blocked(yb).float(), ref, 1e-2, "blocked bf16 convolution vs torch on bf16-rounded operands")


@pytest.mark.parametrize("case", [(6, 64, 32, 128), (5, 128, 16, 256), (9, 64, 8, 64), (3, 64, 64, 64), (17, 256, 8, 512)])
def test_channel_blocked_forms_of_a_discriminator_block_equal_the_nchw_bf16_forms(case):
    """Every launch of a discriminator block with its bf16 tensors channel-blocked ([N][C/8][H][W][8], include/agl.h AGL_CONV_X_BLOCKED /
    _Y_BLOCKED / _MASK_BLOCKED; agl/dtrunk.py) against the same launch on NCHW bf16 tensors: the staged pieces, the products, the order
    of accumulation and the one rounding of the output are the same, so every result must agree BIT FOR BIT —
      forward : c1 3x3 (blocked -> blocked, input ReLU, bias, ReLU), pooled 4x4 / stride 2 (blocked -> fp32 NCHW), 1x1 + fp32 addend
                (fp32 NCHW -> blocked), the shortcut's average pool of a blocked tensor, agl_to_blocked;
      backward: both weight gradients with a blocked x, both input gradients with a blocked ReLU mask, the pool's backward."""
    from agl import lib as L
    N, C, H, Cout = case
    r16 = lambda t: t.to(torch.bfloat16)
    o, w1, b1 = rn(N, C, H, H), rn(C, C, 3, 3, seed=1) * (1.0 / (C * 9) ** 0.5), rn(C, seed=2)
    w4, b2 = rn(Cout, C, 4, 4, seed=3) * (1.0 / (C * 16) ** 0.5), rn(Cout, seed=4)
    wsc, bsc = rn(Cout, C, 1, 1, seed=5) * (1.0 / C ** 0.5), rn(Cout, seed=6)
    d = rn(N, Cout, H // 2, H // 2, seed=7)
    dh_in = rn(N, C, H, H, seed=8)
    with L.conv_flags(L.CONV_BF16 | L.CONV_ANY_GRID):
        od = dev(o)
        o16 = r16(od)
        ob = L.to_blocked_dev(od)
        assert torch.equal(ob, L.to_blocked(od)) and torch.equal(L.to_blocked_dev(o16), ob) and torch.equal(L.from_blocked(ob), o16)
        # forward
        h16 = L.conv2d_fwd(o16, dev(w1), dev(b1), 1, 1, 0, True, True, out_bf16=True)
        hb = L.conv2d_fwd(ob, dev(w1), dev(b1), 1, 1, 0, True, True, out_blk=True)
        assert L.is_blk(hb) and torch.equal(L.from_blocked(hb), h16), "c1: blocked in / out"
        s16, sb = L.avgpool2_fwd(o16, True), L.avgpool2_fwd(ob, True)
        assert torch.equal(s16, sb), "average pool of a blocked tensor"
        hp16 = L.conv2d_fwd(h16, dev(w4), dev(b2), 2, 1)
        hpb = L.conv2d_fwd(hb, dev(w4), dev(b2), 2, 1)
        assert hpb.dtype == torch.float32 and torch.equal(hp16, hpb), "pooled 4x4 / stride 2: blocked x"
        out16 = L.conv2d_fwd_addend(s16, dev(wsc), dev(bsc), hp16, 1, 0, out_bf16=True)
        outb = L.conv2d_fwd_addend(s16, dev(wsc), dev(bsc), hp16, 1, 0, out_blk=True)
        assert L.is_blk(outb) and torch.equal(L.from_blocked(outb), out16), "1x1 + addend: blocked y"
        # backward
        dd, dhd = dev(d), dev(dh_in)
        dh16 = L.conv2d_bwd_data(dd, dev(w4), (H, H), 2, 1, pos_mask=h16)
        dhb = L.conv2d_bwd_data(dd, dev(w4), (H, H), 2, 1, pos_mask=hb)
        assert torch.equal(dh16, dhb), "4x4 / stride-2 input gradient: blocked mask"
        dw4_16, dw4_b = L.conv2d_bwd_weight(dd, h16, 4, 2, 1), L.conv2d_bwd_weight(dd, hb, 4, 2, 1)
        assert torch.equal(dw4_16, dw4_b), "4x4 / stride-2 weight gradient: blocked x"
        dw1_16, dw1_b = L.conv2d_bwd_weight(dhd, o16, 3, 1, 1, in_relu=True), L.conv2d_bwd_weight(dhd, ob, 3, 1, 1, in_relu=True)
        assert torch.equal(dw1_16, dw1_b), "3x3 weight gradient: blocked x, input ReLU"
        ds = dev(rn(N, C, H // 2, H // 2, seed=9))
        do16 = L.avgpool2_bwd(ds, o16, True)
        dob = L.avgpool2_bwd(ds, ob, True)
        assert torch.equal(do16, dob), "pool backward: blocked mask"
        do16 = L.conv2d_bwd_data(dhd, dev(w1), (H, H), 1, 1, pos_mask=o16, out=do16, accumulate=True)
        dob = L.conv2d_bwd_data(dhd, dev(w1), (H, H), 1, 1, pos_mask=ob, out=dob, accumulate=True)
        assert torch.equal(do16, dob), "3x3 input gradient: blocked mask, accumulate"
    # and the chain equals torch on bf16-rounded operands within the bf16 bars
    r = lambda t: t.to(torch.bfloat16).to(torch.float32)
    h_ref = torch.relu(TF.conv2d(torch.relu(r(o)), r(w1), b1, padding=1))
    close(L.from_blocked(hb).float(), h_ref, 1e-2, "c1 vs torch")
    hp_ref = TF.conv2d(r(h_ref), r(w4), b2, stride=2, padding=1)
    close(hpb, hp_ref, 2e-2, "pooled convolution vs torch")


def test_first_block_of_a_flat_discriminator_with_channel_blocked_tensors():
    """The first block of the object / attribute discriminator (3 -> 64 on the crop, no down-sampling) with h and the block output
    channel-blocked: the few-input-channel stream kernel writes blocked pieces, the second convolution reads blocked h and writes the
    blocked sum with the 3-channel shortcut evaluated in its epilogue; h as the blocked x of the weight gradient and as the blocked mask of
    the input gradient.  Bit identity with the NCHW bf16 launches."""
    from agl import lib as L
    N, C, H = 7, 64, 32
    x, w1, b1 = rn(N, 3, H, H), rn(C, 3, 3, 3, seed=1) * 0.2, rn(C, seed=2)
    w2, b2 = rn(C, C, 3, 3, seed=3) * (1.0 / (C * 9) ** 0.5), rn(C, seed=4)
    wsc, bsc = rn(C, 3, seed=5) * 0.5, rn(C, seed=6)
    d = rn(N, C, H, H, seed=7)
    with L.conv_flags(L.CONV_BF16 | L.CONV_ANY_GRID):
        xd = dev(x)
        h16 = L.conv2d_fwd(xd, dev(w1), dev(b1), 1, 1, 0, False, True, out_bf16=True)
        hb = L.conv2d_fwd(xd, dev(w1), dev(b1), 1, 1, 0, False, True, out_blk=True)
        assert L.is_blk(hb) and torch.equal(L.from_blocked(hb), h16), "3 -> C stream kernel: blocked y"
        o16 = L.conv2d_fwd_shortcut(h16, dev(w2), dev(b2), xd, dev(wsc), dev(bsc), 1, out_bf16=True)
        ob = L.conv2d_fwd_shortcut(hb, dev(w2), dev(b2), xd, dev(wsc), dev(bsc), 1, out_blk=True)
        assert L.is_blk(ob) and torch.equal(L.from_blocked(ob), o16), "second convolution + shortcut: blocked x and y"
        dd = dev(d)
        assert torch.equal(L.conv2d_bwd_data(dd, dev(w2), (H, H), 1, 1, pos_mask=h16), L.conv2d_bwd_data(dd, dev(w2), (H, H), 1, 1, pos_mask=hb))
        assert torch.equal(L.conv2d_bwd_weight(dd, h16, 3, 1, 1), L.conv2d_bwd_weight(dd, hb, 3, 1, 1))


@pytest.mark.parametrize("mode", ["bf16", "split3"])
def test_odd_sized_input_gradient_on_concurrent_streams(mode):
    """The training step runs the three generator branches' layout encoders on three streams, so three of these launches (paired-phase
    kernel + phase_edge_k, sharing one cached weight pack) overlap on the chip.  A build of phase_edge_k that read its taps from a
    pre-packed edge table gave different edge rows under this overlap (and only then: alone it matched torch), so the overlap itself is
    tested: 20 rounds of three concurrent calls must reproduce, bit for bit, what each call gives alone."""
    from agl import lib as L
    flags = (L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3)
    w = torch.nn.Parameter(dev(rn(256, 128, 4, 4, seed=1) * 0.05))
    ws = L.WeightSrc(w, lambda: 0)
    dys = [dev(rn(n, 256, 16, 16, seed=3 + i)) for i, n in enumerate((33, 66, 33))]
    with L.conv_flags(flags):
        ref = [L.conv2d_bwd_data(d, w, (33, 33), 2, 1, wsrc=ws) for d in dys]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in dys]
        for it in range(20):
            outs = []
            for d, st in zip(dys, streams):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append(L.conv2d_bwd_data(d, w, (33, 33), 2, 1, wsrc=ws))
            torch.cuda.synchronize()
            for k, (o, r) in enumerate(zip(outs, ref)):
                assert torch.equal(o, r), (it, k, float((o - r).abs().max()))


@pytest.mark.parametrize("mode", ["bf16", "split3"])
@pytest.mark.parametrize("case", [(3, 64, 32, 32, 3), (2, 128, 64, 64, 3), (5, 48, 16, 32, 1), (2, 32, 8, 64, 4)])
def test_few_channel_7x7_as_vertical_conv_plus_diagonal_sum(case, mode):
    """Decoder c4 / c7 (64|128 -> 3, 7x7, generator_obj_att.py:544, generator_obj_att128.py:557) and the input gradients of the
    3-channel first layers (CropEncoder.c1 :374, Decoder.c5): in the matrix-core modes they run as a 7x1 vertical convolution with
    7*CO output planes on the bf16 matrix cores followed by a diagonal sum of shifted planes (csrc/pconv.hip pconv_vert_try).
    Forward with bias / ReLU / accumulate and the input-gradient form with a ReLU mask, against torch (operands rounded to bf16 in
    bf16 mode); in split mode also the fp64 accuracy class (<= 2x the exact fp32 kernel's error)."""
    from agl import lib as L
    N, C, H, W, CO = case
    r = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if mode == "bf16" else (lambda t: t)
    x, w, b = rn(N, C, H, W), rn(CO, C, 7, 7, seed=1) * (1.0 / (C * 49) ** 0.5), rn(CO, seed=2)
    flags = L.CONV_BF16 if mode == "bf16" else L.CONV_SPLIT3
    xd, wd, bd = dev(x), dev(w), dev(b)
    ref = TF.conv2d(r(x), r(w), b, padding=3)
    with L.conv_flags(flags):
        y = L.conv2d_fwd(xd, wd, bd, 1, 3)
        assert L.load().agl_conv2d_last_pipe() == (1 if mode == "bf16" else 3), "the call must run on the matrix cores"
        close(y, ref, 2e-5, "few-output-channel 7x7 forward")
        close(L.conv2d_fwd(xd, wd, bd, 1, 3, relu=True), torch.relu(ref), 2e-5, "with ReLU")
        base = rn(N, CO, H, W, seed=4)
        close(L.conv2d_fwd(xd, wd, None, 1, 3, out=dev(base).clone(), accumulate=True), base + TF.conv2d(r(x), r(w), None, padding=3), 2e-5,
              "accumulating")
        # input gradient of a CO -> C layer (weights (C, CO, 7, 7)): dx has CO channels
        w2 = rn(C, CO, 7, 7, seed=5) * (1.0 / (C * 49) ** 0.5)
        dy = rn(N, C, H, W, seed=6)
        xg = torch.zeros(N, CO, H, W, requires_grad=True)
        TF.conv2d(xg, r(w2), None, padding=3).backward(r(dy))
        mask = rn(N, CO, H, W, seed=7)
        dx = L.conv2d_bwd_data(dev(dy), dev(w2), (H, W), 1, 3, pos_mask=dev(mask))
        assert L.load().agl_conv2d_last_pipe() == (1 if mode == "bf16" else 3)
        close(dx, xg.grad * (mask > 0), 5e-5, "few-input-channel 7x7 input gradient with ReLU mask")
    if mode == "split3":
        y64 = TF.conv2d(x.double(), w.double(), b.double(), padding=3)
        y_exact = L.conv2d_fwd(xd, wd, bd, 1, 3)
        e_split, e_exact = float((y.cpu().double() - y64).abs().max()), float((y_exact.cpu().double() - y64).abs().max())
        assert e_split <= 2.0 * e_exact + 2e-7 * float(y64.abs().max()), (e_split, e_exact)


def test_conv_batchnorm_partials_with_a_large_channel_offset():
    """the convolution epilogue's BatchNorm partials are shift-invariant — every lane accumulates deviations from
    the first value it sees, a partial row is (count, mean, M2), and rows are merged with Chan's update in double — so a channel with
    |mean| >> std costs no accuracy (raw fp32 sum / sum-of-squares partials lost 1e-2 of rstd at mean/std = 100).  With a bias of 100
    or 1000 on unit-variance outputs the fused statistics agree with agl_bn_stats (double accumulation over the stored tensor) to
    1e-4 in rstd and 1e-6 of the mean; the same at mean/std = 3."""
    from agl import lib as L
    N, Cin, H, Cout = 8, 32, 32, 64
    x, w = rn(N, Cin, H, H), rn(Cout, Cin, 4, 4, seed=1) * (1.0 / (Cin * 16) ** 0.5)
    for offset, tol_rstd in ((1000.0, 1e-4), (100.0, 1e-4), (3.0, 1e-4)):
        b = torch.full((Cout,), offset)
        with L.conv_flags(L.CONV_SPLIT3 | L.CONV_ANY_GRID):
            y, part, rows = L.conv2d_fwd_stats(dev(x), dev(w), dev(b), 2, 1)
        assert part is not None and rows > 0
        cnt = y.numel() // Cout
        m1, r1 = L.bn_stats_from_partials(part, rows, Cout, cnt, 1e-5, 0.1)
        m2, r2 = L.bn_stats(y, 1e-5, 0.1)
        std = float((1.0 / r2).mean())
        assert abs(float(m2.mean()) / std) > 0.5 * offset / 1.2, "the case must have the intended mean / std ratio"
        close(m1, m2, 1e-6, f"mean at offset {offset}")
        rel = float(((r1 - r2).abs() / r2).max())
        assert rel <= tol_rstd, (offset, rel)


def test_bn_running_update_replay_single_and_batched():
    """A statistics call leaves its (mean, unbiased variance) in double; agl_bn_running_update re-applies the running-statistics update
    from them bit-identically to the update the statistics call itself makes, and agl_bn_running_update_multi does the same for a
    whole tape in one launch — including a layer that appears twice (its two updates chain in array order) and more items than one
    launch holds."""
    from agl import lib as L
    layers = []
    for k, (N, Cc, H) in enumerate([(6, 64, 8), (3, 130, 4), (5, 32, 16)] * 10):       # 30 items > AGL_BN_UPDATE_MAX
        x = dev(rn(N, Cc, H, H, seed=k) * (1.0 + 0.1 * k) + 0.3 * k)
        rm, rv = dev(rn(Cc, seed=100 + k)), dev(rn(Cc, seed=200 + k).abs() + 0.5)
        nbt = torch.full((1,), 7 + k, dtype=torch.int64, device=DEV)
        layers.append((x, rm, rv, nbt))
    direct, entries, singles = [], [], []
    for x, rm, rv, nbt in layers:
        a = (rm.clone(), rv.clone(), nbt.clone())
        L.bn_stats(x, 1e-5, 0.1, *a)                                    # the statistics call updates a's copies itself
        direct.append(a)
        mom = torch.empty(2 * x.shape[1], dtype=torch.float64, device=DEV)
        L.bn_stats(x, 1e-5, 0.1, None, None, None, moments=mom)         # moments only
        b, c = (rm.clone(), rv.clone(), nbt.clone()), (rm.clone(), rv.clone(), nbt.clone())
        entries.append((mom, *b))
        singles.append(c)
        L.bn_running_update(mom, 0.1, *c)
    L.bn_running_update_many(entries, 0.1)
    for a, e, c in zip(direct, entries, singles):
        for t_direct, t_many, t_single in zip(a, e[1:], c):
            assert torch.equal(t_direct, t_many) and torch.equal(t_direct, t_single)
    # one layer twice in one batch: the second update starts from the first one's result
    x, rm, rv, nbt = layers[0]
    mom = entries[0][0]
    twice = (rm.clone(), rv.clone(), nbt.clone())
    L.bn_running_update_many([(mom, *twice), (mom, *twice)], 0.1)
    ref = (rm.clone(), rv.clone(), nbt.clone())
    L.bn_running_update(mom, 0.1, *ref)
    L.bn_running_update(mom, 0.1, *ref)
    for t1, t2 in zip(twice, ref):
        assert torch.equal(t1, t2)


@pytest.mark.parametrize("shape", [(6, 64, 32, 128), (3, 128, 16, 256), (210, 64, 64, 128), (9, 32, 8, 64), (2, 48, 64, 80)])
def test_box_filtered_map_stored_as_bf16_gives_identical_results(shape):
    """bf16 arithmetic (config 3): the box-filtered map of a down-sampling block is read by one convolution and its weight gradient
    only; written as bf16 (agl_box2_fwd_bf16) and read with AGL_CONV_X_BF16 it holds exactly the values those kernels round the fp32
    map to when they stage it — output, weight gradient and bias gradient must be bit-identical to the fp32-stored path, and the
    tensor must really be bf16 where the matrix-core kernels take the layer."""
    from agl import functional as F, lib as L
    N, Cin, H, Cout = shape
    x, w, b = rn(N, Cin, H, H), rn(Cout, Cin, 3, 3, seed=1) * (1.0 / (Cin * 9) ** 0.5), rn(Cout, seed=2)
    gy = rn(N, Cout, H // 2, H // 2, seed=3)
    res = []
    with L.conv_flags(L.CONV_BF16):
        took = L.box_input_as_bf16(N, Cin, H + 1, H + 1, Cout, True)
        for as16 in (True, False):
            F.BOX_BF16 = as16
            try:
                xg, wg, bg = (dev(t).requires_grad_(True) for t in (x, w, b))
                y = F.conv3x3_avgpool2(xg, wg, bg)
                saved = [t for t in y.grad_fn.saved_tensors if t is not None and t.shape[-1] == H + 1]
                y.backward(dev(gy))
            finally:
                F.BOX_BF16 = True
            res.append((y.detach(), wg.grad, bg.grad, xg.grad, saved[0].dtype))
    assert res[1][4] == torch.float32
    assert res[0][4] == (torch.bfloat16 if took else torch.float32)
    for a_, b_, what in zip(res[0][:4], res[1][:4], ("y", "dw", "db", "dx")):
        assert torch.equal(a_, b_), f"{what} differs between the bf16-stored and the fp32-stored map"
    if shape[0] <= 9:
        r = lambda t: t.to(torch.bfloat16).to(torch.float32)
        xb = TF.avg_pool2d(TF.pad(x, (1, 1, 1, 1)), 2, stride=1)
        close(res[0][0], TF.conv2d(r(xb), r(w), b, stride=2), 3e-5, "y vs an fp32 convolution of the bf16-rounded operands")


@pytest.mark.parametrize("N", [16, 40])
def test_flat_block_first_conv_output_stored_as_bf16_gives_identical_results(N):
    """bf16 arithmetic: in a discriminator block without down-sampling (reference discriminator.py:29-60, 128 px object / attribute
    discriminators) h = relu(c1(x)) is read by c2 only — as its input, as the x operand of its weight gradient and as the ReLU mask
    of its input gradient.  Stored as bf16 (AGL_CONV_Y_BF16 -> AGL_CONV_X_BF16 / AGL_CONV_MASK_BF16) it holds the values c2 rounds it
    to anyway: output and every gradient must be bit-identical to the fp32-stored path."""
    from agl import functional as F, lib as L
    from agl.discriminator import OptimizedBlock
    torch.manual_seed(3)
    blk = OptimizedBlock(3, 64, downsample=False).to(DEV)
    x = rn(N, 3, 64, 64)
    gy = rn(N, 64, 64, 64, seed=4)
    res = []
    with L.conv_flags(L.CONV_BF16):
        took = L.first_conv_output_as_bf16(N, 3, 64, 64, 64, 64, 3, True, True)
        for h16 in (True, False):
            F.H_BF16 = h16
            try:
                for p_ in blk.parameters():
                    p_.grad = None
                xg = dev(x).requires_grad_(True)
                y = blk(xg)
                y.backward(dev(gy))
            finally:
                F.H_BF16 = True
            res.append([y.detach().clone(), xg.grad.clone()] + [p_.grad.clone() for p_ in blk.parameters()])
    assert took, "the 64 -> 64 layer at this size is expected on the matrix-core kernels"
    for k, (a_, b_) in enumerate(zip(res[0], res[1])):
        assert torch.equal(a_, b_), f"tensor {k} differs between the bf16-stored and the fp32-stored h"
