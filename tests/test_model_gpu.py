"""GPU parity of whole networks and of the full G+D training iteration: the HIP path against the CPU
oracle (oracle/graph.py, pinned bit-exact to the imported reference by oracle/make_golden.py) and
against the reference-generated fixtures tests/golden/step64.npz / step128.npz.

Tolerances (fp32, stated per SURVEY.md §8c): network outputs <= 1e-3 relative-to-max, losses <= 1e-4
relative (abs 1e-4 floor), per-tensor gradients <= 5e-3 relative L2 / norm (the CPU oracle's own
fp32-vs-fp64 spread on this network is 1.5e-3 relative L2, measured in the build container, so this is
~3x the reference arithmetic's intrinsic uncertainty), post-step state checksums <= 1e-4 of the tensor's
abs-sum plus an allowance for Adam sign flips: Adam's first updates are lr*sign(g), so an element whose
gradient is below the rounding noise moves by +-lr in either implementation (2*lr per flipped element)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, b, tol, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6)
    assert err <= tol, f"{what}: rel-to-max err {err:.3e} > {tol:.1e}"
    return err


def build_nets(res128):
    from oracle.fill import fill_state
    if res128:
        from models.generator_obj_att128 import Generator
        from models.discriminator import AttributeDiscriminator128 as AttD
    else:
        from models.generator_obj_att import Generator
        from models.discriminator import AttributeDiscriminator as AttD
    from models.discriminator import ImageDiscriminator, ObjectDiscriminator, add_sn
    G = Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=64 if res128 else 32, attribute_dim=106)
    Di = add_sn(ImageDiscriminator(conv_dim=64))
    Do = add_sn(ObjectDiscriminator(n_class=179))
    Da = add_sn(AttD(n_attribute=106))
    nets = [G, Di, Do, Da]
    for m in nets:
        m.load_state_dict(fill_state(m.state_dict()))
        m.to(DEV)
    return nets


def tensors(batch_np):
    return {k: torch.from_numpy(v) for k, v in batch_np.items()}


@pytest.mark.parametrize("res", [64, 128])
def test_generator_vs_oracle(res):
    from agl import synth
    import oracle.graph as OG, oracle.step as OS
    res128 = res == 128
    G = build_nets(res128)[0]
    P = OS.as_params({k: v.cpu() for k, v in G.state_dict().items()})
    bn = synth.make_batch(4, res, seed=5, objs_per_image=[3, 5, 2, 4])
    b = tensors(bn)
    O = b["objs"].shape[0]
    g = torch.Generator().manual_seed(3)
    eps = [torch.randn(O, 64, generator=g) for _ in range(3)]
    out_o = OG.generator(P, b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], b["z"], b["attribute"],
                         b["masks_shift"], b["boxes_shift"], b["attribute_est"], obj_size=G.obj_size, res128=res128,
                         train=True, eps=eps)
    d = {k: (v.to(DEV) if k != "obj_to_img" else v) for k, v in b.items()}
    out_g = G(d["imgs"], d["objs"], d["boxes"], d["masks"], d["obj_to_img"], d["z"], d["attribute"], d["masks_shift"],
              d["boxes_shift"], d["attribute_est"], eps=eps)
    names = ["crops_input", "crops_input_rec", "crops_rand", "crops_shift", "img_rec", "img_rand", "img_shift", "mu",
             "logvar", "z_rand_rec", "z_rand_shift"]
    for n, a, r in zip(names, out_g, out_o):
        close(a, r, 1e-3, n)
    gen = torch.Generator().manual_seed(4)
    cots = [torch.randn(t.shape, generator=gen) / t.numel() ** 0.5 for t in out_o]
    live = [i for i, t in enumerate(out_o) if t.requires_grad]        # crops_input is a crop of the real images
    torch.autograd.backward([out_o[i] for i in live], [cots[i] for i in live])
    torch.autograd.backward([out_g[i] for i in live], [cots[i].to(DEV) for i in live])
    worst = 0.0
    gmax = max(float(P[k].grad.norm()) for k, _ in G.named_parameters())
    for k, p in G.named_parameters():
        go = P[k].grad
        if float(go.norm()) < 1e-6 * gmax:      # mathematically zero (e.g. a bias in front of BatchNorm): pure rounding noise
            assert float(p.grad.norm()) < 1e-4 * gmax, k
            continue
        rel = float((p.grad.cpu() - go).norm() / max(float(go.norm()), 1e-3 * gmax))   # floor: border-only gradients
        worst = max(worst, rel)
        assert rel <= 1e-2, (k, rel)        # whole-network gradients: <= ~7x the oracle's own fp32-vs-fp64 spread
    for k, v in G.state_dict().items():          # BN running statistics after the forward
        if "running" in k:
            close(v, P[k], 1e-4, k)
    print("worst grad rel-L2", worst)


@pytest.mark.parametrize("which", ["img", "obj", "att", "att128"])
def test_discriminator_vs_oracle(which):
    import oracle.graph as OG, oracle.step as OS
    nets = build_nets(which == "att128")
    net = {"img": nets[1], "obj": nets[2], "att": nets[3], "att128": nets[3]}[which]
    P = OS.as_params({k: v.cpu() for k, v in net.state_dict().items()})
    shape = {"img": (3, 3, 64, 64), "obj": (5, 3, 32, 32), "att": (5, 3, 32, 32), "att128": (3, 3, 64, 64)}[which]
    g = torch.Generator().manual_seed(2)
    for call in range(2):                          # two training forwards: SN state must advance identically
        x = torch.randn(*shape, generator=g)
        xo = x.clone().requires_grad_(True)
        xg = x.to(DEV).requires_grad_(True)
        if which == "img":
            yo, yg = [OG.image_discriminator(P, xo)], [net(xg)]
        elif which == "obj":
            yo, yg = list(OG.object_discriminator(P, xo)), list(net(xg))
        else:
            yo, yg = [OG.attribute_discriminator(P, xo, True, which == "att128")], [net(xg)]
        cots = [torch.randn(t.shape, generator=g) for t in yo]
        for v in P.values():
            v.grad = None
        net.zero_grad()
        torch.autograd.backward(yo, cots)
        torch.autograd.backward(yg, [c.to(DEV) for c in cots])
        for a, r in zip(yg, yo):
            close(a, r, 1e-4, f"{which} logits call {call}")
        rel = float((xg.grad.cpu() - xo.grad).norm() / xo.grad.norm())
        assert rel <= 5e-3, (f"{which} dx call {call}", rel)
        for k, p in net.named_parameters():
            go = P[k].grad
            rel = float((p.grad.cpu() - go).norm() / (go.norm() + 1e-12))
            assert rel <= 5e-3, (k, rel)
        for k, v in net.state_dict().items():
            if k.endswith(("weight_u", "weight_v")):
                close(v, P[k], 1e-4, k)


@pytest.mark.parametrize("which", ["img", "obj", "att128"])
def test_block_chain_node_with_channel_blocked_tensors_is_bit_identical(which):
    """agl.dtrunk with its bf16 tensors channel-blocked (D_BLOCKED, the default) against the same node with NCHW bf16 tensors (the
    round-4 form): the launches stage and store different layouts of the same values, so the logits, the input gradient, every parameter
    gradient and the spectral-norm state must agree BIT FOR BIT — at config-3 extents, where every covered block takes the blocked form."""
    import copy
    from agl import dtrunk as T
    from agl import lib as L
    nets = build_nets(which == "att128")
    net = {"img": nets[1], "obj": nets[2], "att128": nets[3]}[which]
    shape = {"img": (32, 3, 128, 128), "obj": (210, 3, 64, 64), "att128": (96, 3, 64, 64)}[which]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g)
    res = {}
    for blocked in (True, False):
        nd = copy.deepcopy(net)
        xg = x.to(DEV).requires_grad_(True)
        prev, T.D_BLOCKED = T.D_BLOCKED, blocked
        T._cover_memo.clear(); T._blocked_memo.clear()
        try:
            with L.conv_flags(L.CONV_BF16):
                yg = [nd(xg)] if which != "obj" else list(nd(xg))
                if blocked:
                    kinds = ["first_down" if which == "img" else "first_flat"] + ["down"] * (len(nd.main) - 1)
                    chans = [(3, 64)] + [(64 << k, 128 << k) for k in range(4)] + ([(1024, 1024)] if which == "att128" else [])
                    k0, k1, o16 = T.cover(kinds, chans, shape[0], shape[2], shape[3])
                    assert T.blocked_ok(kinds, chans, k0, k1, o16, shape[0], shape[2], shape[3]), "the blocked form must be taken at these extents"
                cots = [torch.randn(t.shape, generator=torch.Generator().manual_seed(4)).to(DEV) for t in yg]
                torch.autograd.backward(yg, cots)
        finally:
            T.D_BLOCKED = prev
            T._cover_memo.clear(); T._blocked_memo.clear()
        torch.cuda.synchronize()
        res[blocked] = ([t.detach() for t in yg], xg.grad, {k: q.grad for k, q in nd.named_parameters()},
                        {k: v for k, v in nd.state_dict().items() if k.endswith(("weight_u", "weight_v"))})
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b), ("logits", float((a - b).abs().max()))
    assert torch.equal(res[True][1], res[False][1]), ("input gradient", float((res[True][1] - res[False][1]).abs().max()))
    for k in res[True][2]:
        assert torch.equal(res[True][2][k], res[False][2][k]), ("gradient", k, float((res[True][2][k] - res[False][2][k]).abs().max()))
    for k in res[True][3]:
        assert torch.equal(res[True][3][k], res[False][3][k]), ("spectral-norm state", k)


@pytest.mark.parametrize("which", ["img", "obj", "att128"])
def test_discriminator_block_chain_as_one_node_with_bf16_activations(which):
    """VERDICT r3 item 1 for the discriminators (agl.dtrunk): in bf16 arithmetic a prefix of the block chain runs as ONE autograd node
    whose internal activations — h = relu(c1(.)) of every block and the block outputs that only convolutions and the shortcut's
    average pool read — are stored as bf16; the residual branch's conv3x3 + avg-pool is one 4x4 / stride-2 convolution of the bf16 h
    with the pooled filter, the first block's 3-channel shortcut is evaluated in its second convolution's epilogue.  Checked at sizes
    where the matrix-core kernels run: the node covers blocks and stores bf16 edges; against the fp32 CPU oracle the logits agree to
    2e-2 relative-to-max (measured 3-9e-3), every parameter gradient to 6e-2 relative L2 (measured <= 4.7e-2, the same tensors as on
    the per-op graph) and the input gradient — which passes the ReLU masks of every block, where bf16 rounding flips the entries
    near zero — to 1.5e-1 (measured 0.09-0.11 on BOTH forms); the node must be as close to fp32 as the per-op graph in the same
    arithmetic (AGL_D_TRUNK off), and the spectral-norm state advances identically."""
    import copy
    import oracle.graph as OG, oracle.step as OS
    from agl import dtrunk as T
    from agl import lib as L
    nets = build_nets(which == "att128")
    net = {"img": nets[1], "obj": nets[2], "att128": nets[3]}[which]
    P = OS.as_params({k: v.cpu() for k, v in net.state_dict().items()})
    shape = {"img": (24, 3, 128, 128), "obj": (96, 3, 64, 64), "att128": (96, 3, 64, 64)}[which]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(*shape, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = ([OG.image_discriminator(P, xo)] if which == "img" else list(OG.object_discriminator(P, xo)) if which == "obj"
          else [OG.attribute_discriminator(P, xo, True, True)])
    cots = [torch.randn(t.shape, generator=g) for t in yo]
    torch.autograd.backward(yo, cots)
    res = {}
    for trunk in (True, False):
        nd = copy.deepcopy(net)
        xg = x.to(DEV).requires_grad_(True)
        prev, T.D_TRUNK = T.D_TRUNK, trunk
        T._cover_memo.clear()
        try:
            with L.conv_flags(L.CONV_BF16):
                yg = [nd(xg)] if which != "obj" else list(nd(xg))
                if trunk:
                    kinds = ["first_down" if which == "img" else "first_flat"] + ["down"] * (len(nd.main) - 1)
                    chans = [(3, 64)] + [(64 << k, 128 << k) for k in range(4)] + ([(1024, 1024)] if which == "att128" else [])
                    k0, k1, o16 = T.cover(kinds, chans, shape[0], shape[2], shape[3])
                    print(f"[{which}] node covers blocks {k0}..{k1 - 1}, bf16 block outputs {o16}")
                    assert k1 - k0 >= (1 if which == "img" else 3) and (which == "img" or any(o16)), (k0, k1, o16)
                torch.autograd.backward(yg, [c.to(DEV) for c in cots])
        finally:
            T.D_TRUNK = prev
            T._cover_memo.clear()
        torch.cuda.synchronize()
        res[trunk] = ([t.detach().cpu() for t in yg], xg.grad.cpu(), {k: q.grad.cpu() for k, q in nd.named_parameters()},
                      {k: v.cpu() for k, v in nd.state_dict().items() if k.endswith(("weight_u", "weight_v"))})
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    stats = {}
    for trunk in (True, False):
        ys, dx, grads, uv = res[trunk]
        worst = max((rel(grads[k], P[k].grad), k) for k in grads)
        stats[trunk] = (rel(dx, xo.grad), worst)
        print(f"[{which}, trunk node {trunk}] vs fp32 oracle: logits {max(rel(a, r) for a, r in zip(ys, yo)):.2e}, dx {stats[trunk][0]:.2e}, "
              f"worst parameter gradient {worst[0]:.2e} ({worst[1]})")
    print(f"[{which}] node vs per-op graph (both bf16): dx {rel(res[True][1], res[False][1]):.2e}")
    for trunk in (True, False):
        ys, dx, grads, uv = res[trunk]
        for a, r in zip(ys, yo):
            close(a, r, 2e-2, f"{which} logits (trunk node {trunk})")
        # (the input gradient passes the ReLU masks of every block: bf16 rounding flips masks near zero, see the config-3 test's note)
        assert stats[trunk][0] <= 1.5e-1, (which, trunk, "dx", stats[trunk][0])
        assert stats[trunk][1][0] <= 6e-2, (which, trunk, stats[trunk][1])
        for k, v in uv.items():
            close(v, P[k], 1e-4, k)
    assert stats[True][0] <= 1.3 * stats[False][0] + 1e-2, ("the node's input gradient must be as close to fp32 as the per-op graph's", stats)


def _run_step_fixture(tag, res128, n_steps, golden_dir, conv_dtype="f32"):
    from agl.trainer import Trainer, batch_to_device
    g = np.load(os.path.join(golden_dir, f"step{tag}.npz"), allow_pickle=False)
    G, Di, Do, Da = build_nets(res128)
    for net, key in ((G, "G"), (Di, "D_img"), (Do, "D_obj"), (Da, "D_att")):   # state_dict layout == reference's
        assert list(net.state_dict().keys()) == [str(s) for s in g[f"s0_statenames_{key}"]], key
    tr = Trainer(G, Di, Do, Da, torch.from_numpy(g["pos_weight"]), conv_dtype=conv_dtype)
    batch = {k[len("batch_"):]: g[k] for k in g.files if k.startswith("batch_")}
    b = batch_to_device(batch, DEV)
    nets = {"G": G, "D_img": Di, "D_obj": Do, "D_att": Da}
    for s in range(n_steps):
        p = f"s{s}_"
        # Step 0 starts from bit-identical state and is checked strictly.  Later steps start from a state that went
        # through Adam's lr*sign(g) first update, where every element with |g| below rounding noise moves by +-lr in
        # either implementation: the trajectories separate by ~1e-2 in the images, so later steps are checked at
        # trajectory level here and strictly by test_second_step_lockstep_vs_oracle (oracle restarted from OUR state).
        loose = s > 0
        tol_out, tol_loss, tol_grad = (5e-2, 2e-2, 1e-1) if loose else (1e-3, 1e-4, 5e-3)
        eps_d = [torch.from_numpy(e) for e in g[p + "eps_d"]]
        eps_g = [torch.from_numpy(e) for e in g[p + "eps_g"]]
        norms = {}

        def grab(which):
            def f(t):
                for k in which:
                    norms[k] = np.array([float(q.grad.double().norm()) for q in nets[k].parameters()])
            return f

        tr.on_d_backward, tr.on_g_backward = grab(["D_img", "D_obj", "D_att"]), grab(["G"])
        tr.step(b, eps_d, eps_g)
        tr.finish()
        torch.cuda.synchronize()
        losses = tr.loss_dict()
        for name, ref in zip(g[p + "loss_names"], g[p + "loss_values"]):
            got = losses[str(name)]
            assert abs(got - ref) <= tol_loss * max(1.0, abs(ref)), (s, str(name), got, float(ref))
        names = ["crops_input", "crops_input_rec", "crops_rand", "crops_shift", "img_rec", "img_rand", "img_shift",
                 "mu", "logvar", "z_rand_rec", "z_rand_shift"]
        for n, t in zip(names, tr.last_outputs):
            if n.startswith("crops"):
                close(t[:2], torch.from_numpy(g[p + "out_" + n + "_head"]), tol_out, n)
                ref_sum = g[p + "out_" + n + "_sum"]
                assert abs(float(t.detach().double().abs().sum()) - ref_sum[1]) <= tol_out * ref_sum[1], n
            else:
                close(t, torch.from_numpy(g[p + "out_" + n]), tol_out, n)
        for k in ([] if loose else nets):
            ref = g[p + f"gradnorm_{k}"]
            rel = np.abs(norms[k] - ref) / (ref + 1e-9)
            bad = np.nonzero((rel > tol_grad) & (ref > 1e-3 * ref.max()))[0]
            assert bad.size == 0, (s, k, [(str(g[p + f"gradnames_{k}"][i]), float(norms[k][i]), float(ref[i])) for i in bad[:5]])
        if loose:
            continue
        for k, net in nets.items():
            ref = g[p + f"state_{k}"]
            gn = dict(zip((str(x) for x in g[p + f"gradnames_{k}"]), g[p + f"gradnorm_{k}"]))
            noise = {n for n, v in gn.items() if v <= 1e-6 * max(gn.values())}   # mathematically-zero gradients (a bias
            # in front of BatchNorm): Adam turns their rounding noise into +-lr steps in ANY implementation
            for i, (name, v) in enumerate(net.state_dict().items()):
                if name in noise:
                    continue
                if not v.is_floating_point():
                    assert float(v) == ref[i][0], (k, name)
                    continue
                sm, ab = float(v.double().sum()), float(v.double().abs().sum())
                flips = 2 * 2e-4 * (2 + 2e-3 * v.numel()) * (s + 1) if name.endswith(("weight", "bias", "weight_orig")) else 0.0
                tol = 1e-4 * max(ref[i][1], 1e-3) + flips
                assert abs(ab - ref[i][1]) <= tol, (s, k, name, ab, ref[i][1])
                assert abs(sm - ref[i][0]) <= tol, (s, k, name, sm, ref[i][0])
        sd = G.state_dict()
        if not res128:      # in the 128 px model c4.bias sits in front of a BatchNorm: zero gradient, noise-driven
            close(sd["decoder.c4.bias"], torch.from_numpy(g[p + "G_decoder_c4_bias"]), 1e-3, "c4.bias after Adam")
        close(sd["decoder.spade_3.param_free_norm.running_var"], torch.from_numpy(g[p + "G_spade3_running_var"]), 1e-4, "spade3 rv")
        close(sd["layout_encoder.bn4.bn.running_mean"], torch.from_numpy(g[p + "G_bn4_running_mean"]), 1e-4, "bn4 rm")
        close(Di.state_dict()["classifier.weight_u"], torch.from_numpy(g[p + "Dimg_classifier_u"]), 1e-4, "Dimg u")
        close(Do.state_dict()["main.4.resi.3.weight_u"], torch.from_numpy(g[p + "Dobj_main4_resi3_u"]), 1e-4, "Dobj u")
        close(Da.state_dict()["main.0.resi.0.bias"], torch.from_numpy(g[p + "Datt_main0_resi0_bias"]), 1e-3, "Datt bias")


def test_full_step_64_vs_reference_fixture(golden_dir):
    _run_step_fixture("64", False, 2, golden_dir)


def test_full_step_64_split_products_vs_reference_fixture(golden_dir):
    """The same fixture, same (fp32) tolerances, with conv_dtype="f32x3": fp32 tensors whose products are formed on the bf16
    matrix cores from three bf16 terms per operand (AGL_CONV_SPLIT3, csrc/pconv.hip) wherever that kernel applies."""
    _run_step_fixture("64", False, 1, golden_dir, conv_dtype="f32x3")


def test_full_step_128_split_products_vs_reference_fixture(golden_dir):
    _run_step_fixture("128", True, 1, golden_dir, conv_dtype="f32x3")


def test_full_step_128_vs_reference_fixture(golden_dir):
    _run_step_fixture("128", True, 1, golden_dir)


def test_adam_kernel_matches_torch():
    from agl import lib as L
    n = 100003
    g = torch.Generator().manual_seed(1)
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) * (10.0 ** float(e)) for e in (-3, 0, -6)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], 2e-4, (0.5, 0.999), eps=1e-8)
    pg, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, gr in enumerate(grads):
        pr.grad = gr.clone()
        opt.step()
        L.adam_step(pg, gr.to(DEV), m, v, 2e-4, 0.5, 0.999, 1e-8, i + 1, 1.0)
        assert float((pg.cpu() - pr.detach()).abs().max()) <= 1e-6, i      # a couple of ulps of |p| <= 5
    st = opt.state[pr]
    close(m, st["exp_avg"], 1e-6, "exp_avg")
    close(v, st["exp_avg_sq"], 1e-6, "exp_avg_sq")


def test_second_step_lockstep_vs_oracle():
    """Carry-over of all training state between iterations (weights, BatchNorm running statistics, spectral-norm
    u/v, Adam moments and step count): run one HIP iteration, restart the CPU oracle from OUR post-step state,
    run the second iteration on both and compare strictly."""
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    import oracle.step as OS
    G, Di, Do, Da = build_nets(False)
    pw = torch.from_numpy(synth.make_pos_weight())
    tr = Trainer(G, Di, Do, Da, pw)
    bn = synth.make_batch(3, 64, seed=11, objs_per_image=[2, 4, 3])
    b = batch_to_device(bn, DEV)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(5)
    eps = [[torch.randn(O, 64, generator=gen) for _ in range(3)] for _ in range(4)]
    tr.step(b, eps[0], eps[1])
    tr.finish()
    torch.cuda.synchronize()
    cpu = lambda net: {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ob = OS.OracleBackend(cpu(G), cpu(Di), cpu(Do), cpu(Da), res128=False, obj_size=32)

    def seed_adam(opt, P, flat, offset):
        o = offset
        for t in OS.leaves(P):
            k = t.numel()
            opt.state[t] = {"step": torch.tensor(float(flat.step_count)), "exp_avg": flat.m[o:o + k].cpu().view(t.shape).clone(),
                            "exp_avg_sq": flat.v[o:o + k].cpu().view(t.shape).clone()}
            o += k
        return o

    seed_adam(ob.opt_g, ob.Pg, tr.flat_g, 0)
    o = seed_adam(ob.opt_i, ob.Pi, tr.flat_d, 0)
    o = seed_adam(ob.opt_o, ob.Po, tr.flat_d, o)
    o = seed_adam(ob.opt_a, ob.Pa, tr.flat_d, o)
    assert o == tr.flat_d.n
    bc = {k: torch.from_numpy(v) for k, v in bn.items()}
    l_or, out_or = OS.run_step(ob, bc, pw, eps[2], eps[3])
    tr.step(b, eps[2], eps[3])
    tr.finish()
    torch.cuda.synchronize()
    l_hip = tr.loss_dict()
    for k, ref in l_or.items():
        assert abs(l_hip[k] - ref) <= 1e-4 * max(1.0, abs(ref)), (k, l_hip[k], ref)
    for a, r in zip(tr.last_outputs, out_or):
        close(a, r, 1e-3, "G output, second step")
    for net, P in ((G, ob.Pg), (Di, ob.Pi), (Do, ob.Po), (Da, ob.Pa)):
        for k, v in net.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                close(v, P[k], 2e-4, k)
            elif k.endswith(("weight_u", "weight_v")):
                # the G-step power iterations run on D weights that Adam updated inside this iteration: an element whose
                # gradient is at rounding-noise level moves by +-lr in either implementation (2*lr = 4e-4 apart), which
                # shows up at ~4e-4 relative-to-max in u/v (seen 1 run in 6); a wrong power iteration is an O(1) error
                close(v, P[k], 2e-3, k)
            elif not v.is_floating_point():
                assert int(v) == int(P[k]), k


@pytest.mark.parametrize("tag,res128", [("64", False), ("128", True)])
def test_step_bf16_mode_vs_reference_fixture(tag, res128, golden_dir):
    """BASELINE configs 3/5 compute the convolutions on bf16 MFMA (bf16 operands, fp32 accumulation; statistics,
    spectral norm, losses and Adam stay fp32).  The reference is fp32 only, so parity for this mode is the fp32
    fixture within a looser, stated tolerance: <= 1 % on every loss, and on the generated images <= 5e-2 worst-pixel
    relative-to-max with <= 1e-2 RMS relative-to-max (SURVEY.md §8c suggests 3e-2 / 1 %; the worst single pixel of the
    128 px model measures 3.5e-2 after ~70 bf16-operand convolutions, its RMS error 4e-3)."""
    from agl import lib as L
    from agl.trainer import Trainer, batch_to_device
    g = np.load(os.path.join(golden_dir, f"step{tag}.npz"), allow_pickle=False)
    G, Di, Do, Da = build_nets(res128)
    tr = Trainer(G, Di, Do, Da, torch.from_numpy(g["pos_weight"]), conv_dtype="bf16")
    b = batch_to_device({k[len("batch_"):]: g[k] for k in g.files if k.startswith("batch_")}, DEV)
    tr.step(b, [torch.from_numpy(e) for e in g["s0_eps_d"]], [torch.from_numpy(e) for e in g["s0_eps_g"]])
    tr.finish()
    torch.cuda.synchronize()
    losses = tr.loss_dict()
    assert L.CONV_FLAGS == 0, "the trainer's conv flags must not leak out of step()"
    for name, ref in zip(g["s0_loss_names"], g["s0_loss_values"]):
        got = losses[str(name)]
        assert abs(got - ref) <= 1e-2 * max(1.0, abs(ref)), (str(name), got, float(ref))
    for n, t in zip(["img_rec", "img_rand", "img_shift"], tr.last_outputs[4:7]):
        ref = torch.from_numpy(g["s0_out_" + n])
        close(t, ref, 5e-2, n + " (bf16 mode)")
        rms = float((t.detach().cpu() - ref).pow(2).mean().sqrt() / ref.abs().max())
        assert rms <= 1e-2, (n, rms)


def test_attribute_estimate_on_device():
    """the per-row python loop of train64.py:156-166 as one kernel."""
    from agl import lib as L
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(37, 106, generator=g)
    logits[5, 10] = logits[5, 70] = logits[5].max() + 1        # tie -> lowest index, like torch.argmax
    attr = (torch.rand(37, 106, generator=g) < 0.01).float()
    attr[::3] = 0
    ref = attr.clone()
    for r in range(37):
        if attr[r].sum() == 0:
            ref[r, int(logits[r].argmax())] = 1
    est = L.attr_estimate(logits.to(DEV), attr.to(DEV))
    assert torch.equal(est.cpu(), ref)


@pytest.mark.parametrize("res", [64, 128])
def test_eval_mode_vs_oracle(res):
    """SURVEY §8f N2 (inference path of test64.py:114-198): eval() uses the BatchNorm running statistics and spectral
    norm without a power iteration; outputs must match the oracle with train=False and no state may change."""
    from agl import synth
    import oracle.graph as OG, oracle.step as OS
    res128 = res == 128
    G, Di, Do, Da = build_nets(res128)
    for m in (G, Di, Do, Da):
        m.eval()
    P = OS.as_params({k: v.cpu() for k, v in G.state_dict().items()})
    b = tensors(synth.make_batch(2, res, seed=8, objs_per_image=[3, 2]))
    eps = [torch.randn(5, 64, generator=torch.Generator().manual_seed(1)) for _ in range(3)]
    with torch.no_grad():
        out_o = OG.generator(P, b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], b["z"], b["attribute"],
                             b["masks_shift"], b["boxes_shift"], b["attribute_est"], obj_size=G.obj_size, res128=res128,
                             train=False, eps=eps)
        d = {k: (v.to(DEV) if k != "obj_to_img" else v) for k, v in b.items()}
        before = {k: v.clone() for k, v in G.state_dict().items()}
        out_g = G(d["imgs"], d["objs"], d["boxes"], d["masks"], d["obj_to_img"], d["z"], d["attribute"], d["masks_shift"],
                  d["boxes_shift"], d["attribute_est"], eps=eps)
        for a, r in zip(out_g, out_o):
            close(a, r, 1e-3, "eval generator output")
        for k, v in G.state_dict().items():
            assert torch.equal(v, before[k]), k
        Pd = OS.as_params({k: v.cpu() for k, v in Di.state_dict().items()})
        u_before = Di.state_dict()["classifier.weight_u"].clone()
        close(Di(out_g[4]), OG.image_discriminator(Pd, out_o[4], train=False), 1e-3, "eval D_img")
        assert torch.equal(Di.state_dict()["classifier.weight_u"], u_before)


def test_rasterize_boxes_on_device():
    """masks built in HBM from the boxes equal the host rasteriser of the batch builder (python round,
    slice clipping), for the regular and the shifted boxes (which may leave the image)."""
    from agl import lib as L, synth
    for R in (64, 128):
        b = synth.make_batch(6, R, seed=17)
        for boxes, masks in ((b["boxes"], b["masks"]), (b["boxes_shift"], b["masks_shift"])):
            got = L.rasterize_boxes(torch.from_numpy(boxes).to(DEV), R)
            assert torch.equal(got.cpu(), torch.from_numpy(masks))


@pytest.mark.parametrize("res", [64, 128])
def test_layout_stage1_closed_form_equals_generic_kernels(res):
    """The closed-form first stage of the layout encoder (two-level images, csrc/layout.hip) against the same stage run
    through the generic kernels (materialised c0 output -> CondBN -> ReLU -> c2): outputs, every gradient and the
    BatchNorm running statistics."""
    from agl import synth
    from agl.generator import LayoutEncoder
    from oracle.fill import fill_state
    torch.manual_seed(0)
    enc = LayoutEncoder(z_dim=64, obj_att_dim=64, class_num=179, clstm_layers=3, pool_to_8=(res == 128))
    enc.load_state_dict(fill_state(enc.state_dict()))
    enc = enc.to(DEV)
    b = tensors(synth.make_batch(3, res, seed=4, objs_per_image=[3, 5, 2]))
    O = b["objs"].shape[0]
    g = torch.Generator().manual_seed(2)
    att, z = torch.randn(O, 64, generator=g), torch.randn(O, 64, generator=g)
    results = []
    for closed in (True, False):
        enc.closed_form_stage1 = closed
        enc.zero_grad()
        enc.bn1.bn.running_mean.zero_(); enc.bn1.bn.running_var.fill_(1.0)
        a, zz = att.to(DEV).requires_grad_(True), z.to(DEV).requires_grad_(True)
        v = torch.cat((a, zz), 1)
        u = __import__("agl.functional", fromlist=["x"]).linear(v, enc.c0.weight.view(64, -1))
        F = __import__("agl.functional", fromlist=["x"])
        if closed:
            bn = enc.bn1.bn
            h = F.layout_stage1(u, b["masks"].to(DEV), b["objs"].to(DEV), enc.bn1.embed.weight, enc.c2.weight, bn.running_mean,
                                bn.running_var, bn.num_batches_tracked, True)
        else:
            h = enc.c2(enc.bn1(F.mask_outer(u, b["masks"].to(DEV), 1), b["objs"].to(DEV), relu=True))
        cot = torch.randn(h.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        h.backward(cot)
        results.append((h.detach().clone(), a.grad.clone(), zz.grad.clone(), enc.c2.weight.grad.clone(), enc.c0.weight.grad.clone(),
                        enc.bn1.embed.weight.grad.clone(), enc.bn1.bn.running_mean.clone(), enc.bn1.bn.running_var.clone()))
    names = ["y", "d obj_att", "d z", "d c2.weight", "d c0.weight", "d bn1.embed", "running_mean", "running_var"]
    for n, x, r in zip(names, results[0], results[1]):
        close(x, r, 2e-4 if n.startswith("d") else 2e-5, n)


def test_batched_convlstm_equals_per_call():
    """Generator.batch_clstm runs the ConvLSTM of the three layout-encoder calls as one batched recurrence; outputs,
    gradients and BatchNorm running statistics must equal the call-by-call schedule (fp32 rounding only: the GEMM
    splits differ with the batch)."""
    from agl import synth
    from agl.trainer import batch_to_device
    bn = synth.make_batch(5, 64, seed=17, objs_per_image=[3, 1, 4, 2, 5])
    b = batch_to_device(bn, DEV)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(3)
    eps = [torch.randn(O, 64, generator=gen) for _ in range(3)]
    res = []
    for flag in (True, False):
        G = build_nets(False)[0]
        G.batch_clstm = flag
        out = G(b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], b["z"], b["attribute"], b["masks_shift"],
                b["boxes_shift"], b["attribute_est"], eps=eps)
        loss = sum((o * o).mean() for o in out[4:7]) + out[9].sum() * 0.01
        loss.backward()
        res.append(([o.detach().clone() for o in out], {k: v.grad.detach().clone() for k, v in G.named_parameters() if v.grad is not None},
                    {k: v.detach().clone() for k, v in G.state_dict().items()}))
    (o1, g1, s1), (o2, g2, s2) = res
    for a, r in zip(o1, o2):
        close(a, r, 2e-5, "output")
    assert g1.keys() == g2.keys()
    for k in g1:
        a, r = g1[k].double(), g2[k].double()
        # fp32 spread of this net: 1.5e-3; biases in front of a BatchNorm have a mathematically zero gradient (noise ~1e-8)
        assert float((a - r).norm()) <= 2e-3 * float(r.norm()) + 1e-6, k
    for k in s1:
        if k.endswith(("running_mean", "running_var")):
            close(s1[k], s2[k], 1e-5, k)
        elif k.endswith("num_batches_tracked"):
            assert int(s1[k]) == int(s2[k]), k


def test_generator_pass_reuse_equals_two_full_passes():
    """Trainer(reuse_generator_pass=True) evaluates the draw-independent generator parts once per iteration and
    replays the BatchNorm statistics of the reused layers; it must match the plain schedule (two full generator
    passes).  After the FIRST iteration both schedules ran on identical weights: losses and every BatchNorm running
    statistic / counter must agree to rounding.  After the second, weights differ by Adam sign flips of noise-level
    gradients (2*lr per element and step), so the comparison is at that granularity."""
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    pw = torch.from_numpy(synth.make_pos_weight())
    bn = synth.make_batch(4, 64, seed=31, objs_per_image=[3, 2, 4, 1])
    b = batch_to_device(bn, DEV)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(12)
    eps = [[torch.randn(O, 64, generator=gen) for _ in range(3)] for _ in range(4)]
    runs = []
    for flag in (True, False):
        nets = build_nets(False)
        tr = Trainer(*nets, pw, reuse_generator_pass=flag)
        snaps = []
        for it in range(2):
            tr.step(b, eps[2 * it], eps[2 * it + 1])
            tr.finish()
            snaps.append((tr.loss_dict(), [{k: v.detach().cpu().clone() for k, v in n.state_dict().items()} for n in nets]))
        runs.append(snaps)
    for it, (stat_tol, w_tol) in enumerate(((1e-5, 4.1e-4), (2e-3, 8.2e-4))):
        (l1, s1), (l2, s2) = runs[0][it], runs[1][it]
        for k, v in l2.items():
            assert abs(l1[k] - v) <= (2e-5 if it == 0 else 2e-4) * max(1.0, abs(v)), (it, k, l1[k], v)
        for a, r in zip(s1, s2):
            for k in r:
                if k.endswith("num_batches_tracked"):
                    assert int(a[k]) == int(r[k]), (it, k)
                elif k.endswith(("running_mean", "running_var", "weight_u", "weight_v")):
                    close(a[k], r[k], stat_tol, f"iteration {it}: {k}")
                else:
                    assert float((a[k].double() - r[k].double()).abs().max()) <= w_tol + 1e-5 * float(r[k].abs().max()), (it, k)


@pytest.mark.parametrize("tag", ["64", "128"])
def test_eval_mode_vs_reference_fixture(tag, golden_dir):
    """N2 pinned to the reference: tests/golden/eval{64,128}.npz hold the outputs of the REFERENCE's netG.eval() and
    netD_*.eval() forwards (test64.py:96-101,132-141; generated by oracle/make_golden.py::eval_mode from closed-form
    weights plus the recorded, power-iterated spectral-norm u/v).  Tolerance 1e-3 relative-to-max per tensor."""
    g = np.load(os.path.join(golden_dir, f"eval{tag}.npz"))
    res128 = tag == "128"
    nets = dict(zip(("G", "D_img", "D_obj", "D_att"), build_nets(res128)))
    for k, m in nets.items():
        sd = m.state_dict()
        for name in sd:
            key = f"sn_{k}_{name}"
            if key in g.files:
                sd[name].copy_(torch.from_numpy(g[key]))
        m.eval()
    G, Di, Do, Da = nets["G"], nets["D_img"], nets["D_obj"], nets["D_att"]
    b = {k[len("batch_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("batch_")}
    d = {k: (v.to(DEV) if k != "obj_to_img" else v) for k, v in b.items()}
    eps = [torch.from_numpy(e) for e in g["eps"]]
    before = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in nets.items()}
    with torch.no_grad():
        out = G(d["imgs"], d["objs"], d["boxes"], d["masks"], d["obj_to_img"], d["z"], d["attribute"], d["masks_shift"],
                d["boxes_shift"], d["attribute_est"], eps=eps)
        names = ["crops_input", "crops_input_rec", "crops_rand", "crops_shift", "img_rec", "img_rand", "img_shift", "mu", "logvar",
                 "z_rand_rec", "z_rand_shift"]
        for n, a in zip(names, out):
            close(a, torch.from_numpy(g["out_" + n]), 1e-3, "eval G " + n)
        img, crops = torch.from_numpy(g["out_img_rand"]).to(DEV), torch.from_numpy(g["out_crops_rand"]).to(DEV)
        close(Di(img), torch.from_numpy(g["d_img"]), 1e-3, "eval D_img")
        src, cls = Do(crops, d["objs"])
        close(src, torch.from_numpy(g["d_obj_src"]), 1e-3, "eval D_obj src")
        close(cls, torch.from_numpy(g["d_obj_cls"]), 1e-3, "eval D_obj cls")
        close(Da(crops), torch.from_numpy(g["d_att"]), 1e-3, "eval D_att")
    for k, m in nets.items():                       # eval forwards must not move any state
        for n, v in m.state_dict().items():
            assert torch.equal(v, before[k][n]), (k, n)


def test_full_step_at_config2_size_vs_oracle():
    """The WHOLE iteration at BASELINE config 2 size (64 px, batch 64, P ~ U{3..9}, O ~ 390) against the CPU oracle on the
    box's host cores: the size-dependent kernel choices (matrix-core kernels only on grids that fill the chip, position-major
    ConvLSTM path needs >= 96 objects, 256x128 tiles, split-K plans, batched layout-encoder calls) are only met end-to-end,
    with BatchNorm / spectral-norm state, at this size.  Run in BOTH fp32 arithmetics against one oracle evaluation: exact fp32
    MFMA, and the split-product mode bench.py measures by default (AGL_CONV_SPLIT3) at the SAME tolerances.  Checked: the 15
    logged losses (<= 1e-4 relative for the D losses computed from identical state, 5e-3 for the G losses that follow the
    discriminators' first lr*sign(g) Adam update), the generated images / latents (<= 2e-3 relative-to-max), every
    per-tensor gradient norm (<= 1e-2 relative, tensors above 1e-3 of the largest norm) AND every gradient tensor's direction:
    relative L2 distance to the oracle's gradient <= 5e-3 for the discriminators (identical state), <= 3e-2 for the generator
    (its gradient flows through the discriminators after their first Adam update), same set of tensors."""
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    import oracle.step as OS
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    pw = torch.from_numpy(synth.make_pos_weight())
    bn = synth.make_batch(64, 64, seed=1234)
    O = bn["objs"].shape[0]
    assert O >= 96 * 3
    gen = torch.Generator().manual_seed(21)
    eps_d = [torch.randn(O, 64, generator=gen) for _ in range(3)]
    eps_g = [torch.randn(O, 64, generator=gen) for _ in range(3)]
    ref = out_ref = None
    ref_norms, ref_grads = {}, {}
    for conv_dtype in ("f32", "f32x3"):
        G, Di, Do, Da = build_nets(False)
        nets = {"G": G, "D_img": Di, "D_obj": Do, "D_att": Da}
        norms, dist = {}, {}
        if ref is None:
            cpu = lambda net: {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
            ob = OS.OracleBackend(cpu(G), cpu(Di), cpu(Do), cpu(Da), res128=False, obj_size=32)

            def grab_ref(which):
                def f(be):
                    st = be.states()
                    for k in which:
                        ref_norms[k] = np.array([float(v.grad.double().norm()) for v in st[k].values() if v.requires_grad])
                        ref_grads[k] = [v.grad.detach().clone() for v in st[k].values() if v.requires_grad]
                return f

            bc = {k: torch.from_numpy(v) for k, v in bn.items()}
            ref, out_ref = OS.run_step(ob, bc, pw, eps_d, eps_g, on_d_backward=grab_ref(["D_img", "D_obj", "D_att"]),
                                       on_g_backward=grab_ref(["G"]))

        def grab(which):
            def f(t):
                for k in which:
                    norms[k] = np.array([float(q.grad.double().norm()) for q in nets[k].parameters()])
                    dist[k] = np.array([float((q.grad.detach().cpu().double() - r.double()).norm())
                                        for q, r in zip(nets[k].parameters(), ref_grads[k])])
            return f

        tr = Trainer(G, Di, Do, Da, pw, conv_dtype=conv_dtype)
        tr.on_d_backward, tr.on_g_backward = grab(["D_img", "D_obj", "D_att"]), grab(["G"])
        tr.step(batch_to_device(bn, DEV), eps_d, eps_g)
        tr.finish()
        torch.cuda.synchronize()
        hip = tr.loss_dict()
        for k, r in ref.items():
            tol = 1e-4 if k.startswith("D/") else 5e-3
            assert abs(hip[k] - r) <= tol * max(1.0, abs(r)), (conv_dtype, k, hip[k], r)
        for i, (a, r) in enumerate(zip(tr.last_outputs, out_ref)):
            close(a, r, 2e-3, f"G output {i} at config-2 size ({conv_dtype})")
        for k in nets:
            rel = np.abs(norms[k] - ref_norms[k]) / (ref_norms[k] + 1e-9)
            bad = np.nonzero((rel > 1e-2) & (ref_norms[k] > 1e-3 * ref_norms[k].max()))[0]
            names = [n for n, _ in nets[k].named_parameters()]
            assert bad.size == 0, (conv_dtype, k, [(names[i], float(norms[k][i]), float(ref_norms[k][i])) for i in bad[:5]])
            rel2 = dist[k] / (ref_norms[k] + 1e-30)
            big = ref_norms[k] > 1e-3 * ref_norms[k].max()
            lim = 3e-2 if k == "G" else 5e-3
            bad = np.nonzero((rel2 > lim) & big)[0]
            print(f"[config-2 size, {conv_dtype}] {k}: worst relative L2 gradient distance {float(rel2[big].max()):.2e} (limit {lim:.0e})")
            assert bad.size == 0, (conv_dtype, k, [(names[i], float(rel2[i])) for i in bad[:5]])


def test_step_128_bf16_at_matrix_core_sizes_vs_oracle():
    """BASELINE config 3 arithmetic (128 px, bf16 MFMA operands) at a size where the image- and object-level convolutions
    run on the matrix-core kernels (the fixture batch is below their occupancy threshold almost everywhere): batch 8,
    against the fp32 CPU oracle within the tolerances stated for the bf16 mode (losses <= 1 %, images <= 5e-2 worst pixel
    and <= 1e-2 RMS, both relative to the image maximum)."""
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    import oracle.step as OS
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    G, Di, Do, Da = build_nets(True)
    cpu = lambda net: {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ob = OS.OracleBackend(cpu(G), cpu(Di), cpu(Do), cpu(Da), res128=True, obj_size=64)
    pw = torch.from_numpy(synth.make_pos_weight())
    bn = synth.make_batch(8, 128, seed=77)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(5)
    eps_d = [torch.randn(O, 64, generator=gen) for _ in range(3)]
    eps_g = [torch.randn(O, 64, generator=gen) for _ in range(3)]
    tr = Trainer(G, Di, Do, Da, pw, conv_dtype="bf16")
    tr.step(batch_to_device(bn, DEV), eps_d, eps_g)
    tr.finish()
    torch.cuda.synchronize()
    hip = tr.loss_dict()
    bc = {k: torch.from_numpy(v) for k, v in bn.items()}
    ref, out_ref = OS.run_step(ob, bc, pw, eps_d, eps_g)
    for k, r in ref.items():
        assert abs(hip[k] - r) <= 1e-2 * max(1.0, abs(r)), (k, hip[k], r)
    for n, t, r in zip(["img_rec", "img_rand", "img_shift"], tr.last_outputs[4:7], out_ref[4:7]):
        close(t, r, 5e-2, n + " (bf16 mode, 128 px)")
        rms = float((t.detach().cpu() - r).pow(2).mean().sqrt() / r.abs().max())
        assert rms <= 1e-2, (n, rms)


def test_full_step_at_config3_size_bf16_vs_oracle():
    """The WHOLE iteration at BASELINE config 3 size — 128 px, batch 32 (O ~ 200 objects of 64x64), bf16 MFMA operands — against
    the fp32 CPU oracle on the box's host cores.  The size-dependent kernel choices (200-workgroup threshold of the matrix-core
    kernels, reduction splits, 256-pixel tiles, packed-weight reuse) differ from the batch-2 fixture and the batch-8 test, so
    the arithmetic mode the bench's 128 px line is measured in is checked at exactly that size: losses <= 1 % relative, images
    <= 5e-2 worst pixel and <= 1e-2 RMS relative to the image maximum, ALL eleven generator outputs (crops <= 5e-2 like the images
    they are cut from; mu / logvar / the two re-encoded latents <= 3e-2 relative-to-max), every large per-tensor gradient norm
    within 10 %, and every large gradient tensor's DIRECTION: relative L2 distance to the oracle's gradient <= 3e-2 for the
    discriminators (measured 0.7-1.7e-2: operands rounded to bf16, 2^-9 relative each, thousands of terms per output) and <= 0.6 for
    the generator (measured 0.21 at the output layer c7 up to 0.47 at the layers furthest from it).  The generator bar is what bf16
    operands cost on THIS graph, not slack for a bug: its loss has kinks everywhere (L1 terms, ~70 ReLU layers), a perturbation of
    relative size e flips a share ~e of the masks / signs and each flip changes its element's gradient by O(1), so the gradient
    distance grows like sqrt(e) — tools/grad_profile.py measures 2-5e-3 between the two fp32-accurate arithmetics of this build
    (exact fp32 MFMA vs split products, e ~ 1e-7) and 0.2-0.5 between bf16 and either (e ~ 4e-3: the same sqrt law), with the
    discriminators' Adam update taken out of the comparison in both (profiles/r04_grad_profile_*.txt).  The oracle's D update is
    replaced by the HIP run's D weights for the same reason (lr * sign(g) flips)."""
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    import oracle.step as OS
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    G, Di, Do, Da = build_nets(True)
    nets = {"G": G, "D_img": Di, "D_obj": Do, "D_att": Da}
    cpu = lambda net: {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    ob = OS.OracleBackend(cpu(G), cpu(Di), cpu(Do), cpu(Da), res128=True, obj_size=64)
    ob16 = OS.OracleBackend(cpu(G), cpu(Di), cpu(Do), cpu(Da), res128=True, obj_size=64)      # (same initial state: the bf16-operand yardstick below)
    pw = torch.from_numpy(synth.make_pos_weight())
    bn = synth.make_batch(32, 128, seed=1234)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(9)
    eps_d = [torch.randn(O, 64, generator=gen) for _ in range(3)]
    eps_g = [torch.randn(O, 64, generator=gen) for _ in range(3)]
    norms, ref_norms, grads, ref_grads = {}, {}, {}, {}

    def grab(which):
        def f(t):
            for k in which:
                norms[k] = np.array([float(q.grad.double().norm()) for q in nets[k].parameters()])
                grads[k] = [q.grad.detach().cpu().clone() for q in nets[k].parameters()]
        return f

    def grab_ref(which):
        def f(be):
            st = be.states()
            for k in which:
                ref_norms[k] = np.array([float(v.grad.double().norm()) for v in st[k].values() if v.requires_grad])
                ref_grads[k] = [v.grad.detach().clone() for v in st[k].values() if v.requires_grad]
        return f

    tr = Trainer(G, Di, Do, Da, pw, conv_dtype="bf16")
    tr.on_d_backward, tr.on_g_backward = grab(["D_img", "D_obj", "D_att"]), grab(["G"])
    tr.step(batch_to_device(bn, DEV), eps_d, eps_g)
    tr.finish()
    torch.cuda.synchronize()
    hip = tr.loss_dict()
    bc = {k: torch.from_numpy(v) for k, v in bn.items()}

    # The discriminators' first Adam update is lr * sign(g): where bf16 noise flips the sign of a small gradient the two
    # implementations' weights end up 2 lr apart, and the generator's gradient (which flows through the updated discriminators)
    # would measure THAT instead of the kernels.  The oracle's D update is therefore replaced by a copy of the weights the HIP
    # iteration used in its G step (its own D gradients are compared before that, from identical state).
    def step_d_synced():
        for P, net in ((ob.Pi, Di), (ob.Po, Do), (ob.Pa, Da)):
            for name, q in net.named_parameters():
                P[name].data.copy_(q.detach().cpu())
    ob.step_d = step_d_synced
    ref, out_ref = OS.run_step(ob, bc, pw, eps_d, eps_g, on_d_backward=grab_ref(["D_img", "D_obj", "D_att"]),
                               on_g_backward=grab_ref(["G"]))
    for k, r in ref.items():
        assert abs(hip[k] - r) <= 1e-2 * max(1.0, abs(r)), (k, hip[k], r)
    for n, t, r in zip(["img_rec", "img_rand", "img_shift"], tr.last_outputs[4:7], out_ref[4:7]):
        close(t, r, 5e-2, n + " (bf16 mode, 128 px, batch 32)")
        rms = float((t.detach().cpu() - r).pow(2).mean().sqrt() / r.abs().max())
        assert rms <= 1e-2, (n, rms)
    out_names = ["crops_input", "crops_input_rec", "crops_rand", "crops_shift", "img_rec", "img_rand", "img_shift", "mu", "logvar",
                 "z_rand_rec", "z_rand_shift"]
    for n, t, r in zip(out_names, tr.last_outputs, out_ref):
        tol = 5e-2 if (n.startswith("crops") or n.startswith("img")) else 3e-2
        err = float((t.detach().cpu().double() - r.double()).abs().max() / max(float(r.abs().max()), 1e-6))
        print(f"[config-3 size, bf16] output {n}: rel-to-max error {err:.2e} (limit {tol:.0e})")
        close(t, r, tol, n + " (bf16 mode, 128 px, batch 32)")
    for k in nets:
        rel = np.abs(norms[k] - ref_norms[k]) / (ref_norms[k] + 1e-9)
        big = ref_norms[k] > 1e-2 * ref_norms[k].max()
        names = [n for n, _ in nets[k].named_parameters()]
        # (norms of bf16-mode gradients against the fp32 oracle: samples of the rounding noise — seen 0.06 .. 0.117 as kernels changed)
        bad = np.nonzero((rel > 0.15) & big)[0]
        print(f"[config-3 size, bf16] {k}: worst relative gradient-norm deviation {float(rel[big].max()):.2e}")
        assert bad.size == 0, (k, [(names[i], float(norms[k][i]), float(ref_norms[k][i])) for i in bad[:5]])
    worst = {}
    for k in nets:
        names = [n for n, _ in nets[k].named_parameters()]
        big = ref_norms[k] > 1e-2 * ref_norms[k].max()
        dist = np.array([float((a.double() - r.double()).norm()) for a, r in zip(grads[k], ref_grads[k])])
        rel2 = dist / (ref_norms[k] + 1e-30)
        order = np.argsort(-(rel2 * big))
        worst[k] = (float(rel2[big].max()), [(names[i], round(float(rel2[i]), 4)) for i in order[:4]])
        print(f"[config-3 size, bf16] {k}: worst relative L2 gradient distance {worst[k][0]:.2e}  {worst[k][1]}")
    for k in nets:
        lim = 0.6 if k == "G" else 3e-2
        assert worst[k][0] <= lim, (k, worst[k])

    # ---- the same iteration against the oracle with bf16-ROUNDED CONVOLUTION OPERANDS (oracle.graph.OPERAND_ROUND: x, w in the forward,
    # dy, w / dy, x in the two gradients, fp32 accumulation — where the HIP bf16 mode rounds).  The comparison above states what the
    # mode costs against the reference's fp32 arithmetic; this one asks whether the bf16-mode kernels (bf16-stored SPADE outputs and
    # trunk activations, the pooled 4x4 filter gradient, bf16 dy operands, the few-channel role swap) compute THAT arithmetic
    # correctly (VERDICT r4 weak 1a / ADVICE r4).  MEASURED RESULT (gpurun_out/r5/t7.log, recorded in DESIGN.md): this yardstick is NOT
    # closer to the HIP run than the fp32 oracle is — generator gradients 0.465 (0.467 vs fp32), D_img 2.9e-2 (1.8e-2), D_obj 8.3e-3
    # (6.6e-3), D_att 1.9e-2 (1.7e-2), images 2.4-3.4e-2 either way.  Two bf16 evaluations that round at slightly different places
    # (exact-fp32 layers below the matrix-core grid threshold, r(W) / sigma vs r(W / sigma), bf16-stored activations, summation order)
    # are two independent perturbations of relative size e ~ 4e-3 of a loss with kinks everywhere, and each is ~sqrt(e) away from the
    # other exactly as it is from the fp32 result.  So the bars below are the ones above (they document that the bf16 yardstick agrees
    # as well as the fp32 one, not better), and the tight fidelity check of the bf16-only code paths is HIP against HIP:
    # test_generator_bf16_only_paths_equal_plain_bf16_arithmetic (fold / bf16-stored SPADE outputs on and off: same arithmetic) and
    # test_discriminator_block_chain_as_one_node_with_bf16_activations.
    import oracle.graph as OG
    ref_norms16, ref_grads16 = {}, {}

    def grab_ref16(which):
        def f(be):
            st = be.states()
            for k in which:
                ref_norms16[k] = np.array([float(v.grad.double().norm()) for v in st[k].values() if v.requires_grad])
                ref_grads16[k] = [v.grad.detach().clone() for v in st[k].values() if v.requires_grad]
        return f

    def step_d_synced16():
        for P, net in ((ob16.Pi, Di), (ob16.Po, Do), (ob16.Pa, Da)):
            for name, q in net.named_parameters():
                P[name].data.copy_(q.detach().cpu())
    ob16.step_d = step_d_synced16
    OG.OPERAND_ROUND = lambda t: t.to(torch.bfloat16).to(torch.float32)
    try:
        ref16, out_ref16 = OS.run_step(ob16, bc, pw, eps_d, eps_g, on_d_backward=grab_ref16(["D_img", "D_obj", "D_att"]),
                                       on_g_backward=grab_ref16(["G"]))
    finally:
        OG.OPERAND_ROUND = None
    for k, r in ref16.items():
        print(f"[config-3 size, bf16 vs bf16-operand oracle] loss {k}: {abs(hip[k] - r) / max(1.0, abs(r)):.2e} (vs fp32 oracle {abs(hip[k] - ref[k]) / max(1.0, abs(ref[k])):.2e})")
        assert abs(hip[k] - r) <= 5e-3 * max(1.0, abs(r)), ("bf16-operand oracle", k, hip[k], r)
    for n, t, r in zip(out_names, tr.last_outputs, out_ref16):
        err = float((t.detach().cpu().double() - r.double()).abs().max() / max(float(r.abs().max()), 1e-6))
        print(f"[config-3 size, bf16 vs bf16-operand oracle] output {n}: rel-to-max error {err:.2e}")
        assert err <= (5e-2 if (n.startswith("crops") or n.startswith("img")) else 3e-2), ("bf16-operand oracle", n, err)
    for k in nets:
        names = [n for n, _ in nets[k].named_parameters()]
        big = ref_norms16[k] > 1e-2 * ref_norms16[k].max()
        dist = np.array([float((a.double() - r.double()).norm()) for a, r in zip(grads[k], ref_grads16[k])])
        rel2 = dist / (ref_norms16[k] + 1e-30)
        order = np.argsort(-(rel2 * big))
        w16 = float(rel2[big].max())
        print(f"[config-3 size, bf16 vs bf16-operand oracle] {k}: worst relative L2 gradient distance {w16:.2e}  "
              f"{[(names[i], round(float(rel2[i]), 4)) for i in order[:4]]}  (vs the fp32 oracle: {worst[k][0]:.2e})")
        assert w16 <= (0.6 if k == "G" else 5e-2), ("bf16-operand oracle", k, w16)


def test_generator_bf16_only_paths_equal_plain_bf16_arithmetic():
    """Fidelity of the code paths that exist only in bf16 mode, HIP against HIP at network level, one 128 px iteration:

    (A) SPADE outputs stored as bf16 inside their consumer's node (agl.functional.SPADE_Y16, the default) against fp32-stored modulated
        tensors: every reader rounds to bf16 when it stages the tensor anyway, so the WHOLE iteration must be BIT-IDENTICAL — all losses,
        all eleven generator outputs, every gradient tensor of all four networks.  This is the tight check of _SpadeThenConv's bf16 y,
        the bf16 dy operand of the transposed convolutions' weight gradients (pbww_k AUX), the bf16 x of the 7x7 few-channel layers and
        their role swap: a defect in any of them breaks equality.
    (B) the BatchNorm / ConditionalBN apply folded into its consumer's staging pass (NORM_FOLD) against the stand-alone apply pass.
        Per layer the two differ in ~1 % of the outputs by one bf16 rounding (tools/fold_diag.py: rms 1e-5 of the output's rms; the
        transform itself agrees to 5e-7) — but bf16 rounding is discontinuous, and a perturbation of ANY size is amplified to the bf16
        noise floor within three or four layers (1e-5 -> 0.25 % of the next layer's operands flip by one 2^-8 step -> 1e-4 -> ...), which
        is also why an oracle with bf16-rounded operands is no closer to this path than the fp32 oracle (the config-3 test above).  So
        (B) can only assert the floor: D-step losses 4e-4, the crop encoder's mu / logvar (four folded layers deep) within 5e-3 of their
        maximum, images within the distance bf16 has from the fp32-accurate arithmetic (max 6e-2, rms 8e-3 of the maximum).  The tight
        statement about the fold is the per-layer one (tests/test_ops_gpu.py: folded vs two passes, both arithmetic modes)."""
    from agl import functional as F
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    pw = torch.from_numpy(synth.make_pos_weight())
    bn = synth.make_batch(8, 128, seed=77)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(5)
    eps = [torch.randn(O, 64, generator=gen) for _ in range(6)]
    old = (F.NORM_FOLD, F.SPADE_Y16)

    def run(fold, y16):
        F.NORM_FOLD, F.SPADE_Y16 = fold, y16
        nets = build_nets(True)
        # (one stream, program order: gradient slots that several branches add to receive their contributions in a fixed order, so that
        #  everything below is reproducible bit for bit — with the concurrent schedule a few slots are summed in the order the streams
        #  finish, and one sample of that order is no yardstick for another: seen 4.5x and 8.5x apart)
        tr = Trainer(*nets, pw, conv_dtype="bf16", streams=False)
        grads = {}

        def grab(tag, which):
            def f(t):
                grads[tag] = [(n, q.grad.detach().cpu().clone()) for net in which for n, q in net.named_parameters()]
            return f
        tr.on_d_backward, tr.on_g_backward = grab("D", nets[1:]), grab("G", nets[:1])
        tr.step(batch_to_device(bn, DEV), eps[:3], eps[3:])
        tr.finish()
        torch.cuda.synchronize()
        return tr.loss_dict(), [t.detach().cpu() for t in tr.last_outputs], grads
    try:
        base = run(True, True)
        again = run(True, True)
        y16_off = run(True, False)
        fold_off = run(False, True)
    finally:
        F.NORM_FOLD, F.SPADE_Y16 = old
    # (A) bit identity — for everything that is bit-reproducible from run to run (anything that is not — none expected on one stream — is
    # held to that run-to-run distance instead)
    assert base[0] == y16_off[0], "losses differ with fp32-stored SPADE outputs"
    for i, (a, b) in enumerate(zip(base[1], y16_off[1])):
        assert torch.equal(a, b), ("generator output", i)
    exact = loose = 0
    for tag in ("D", "G"):
        for (n, a), (_, a2), (_, b) in zip(base[2][tag], again[2][tag], y16_off[2][tag]):
            if n in ("decoder.c4.weight", "decoder.c7.weight"):
                # the 64|128 -> 3 7x7 layers: their weight gradient runs on the exact-fp32 few-channel kernel (few_bww_k, role swap), which
                # does not round its operands — it reads the modulated tensor as stored, bf16 or fp32: one bf16 rounding of one operand apart
                assert float((a.double() - b.double()).norm()) <= 1e-2 * float(a.double().norm()), ("gradient", tag, n)
            elif torch.equal(a, a2):
                assert torch.equal(a, b), ("gradient", tag, n, float((a - b).abs().max()))
                exact += 1
            else:
                nrm = float(a.double().norm()) + 1e-30
                # (one sample of an order-dependent fp32 sum is a noisy yardstick — seen 4.5x once: eight times it, or 3e-4, an order below
                #  what ONE bf16 rounding of an operand does to a gradient, 4e-3)
                d_ab, d_aa = float((a.double() - b.double()).norm()) / nrm, float((a.double() - a2.double()).norm()) / nrm
                assert d_ab <= max(8 * d_aa, 3e-4), ("gradient", tag, n, d_ab, d_aa)
                loose += 1
    print(f"[SPADE outputs bf16 / fp32 stored] gradient tensors bit-identical: {exact}, order-dependent from run to run: {loose}")
    assert exact >= 50, (exact, loose)      # (the discriminators' slots and the single-branch generator layers are reproducible bit for bit)
    # (B) the bf16 floor
    la, lb = base[0], fold_off[0]
    for k in la:
        if k.startswith("D/"):
            # (a floor, not a tolerance: the two runs are two samples of the bf16 rounding noise — seen 0.2e-4 .. 1.3e-4 as kernels changed)
            assert abs(la[k] - lb[k]) <= 4e-4 * max(1.0, abs(lb[k])), (k, la[k], lb[k])
    names = ["crops_input", "crops_input_rec", "crops_rand", "crops_shift", "img_rec", "img_rand", "img_shift", "mu", "logvar", "z_rand_rec", "z_rand_shift"]
    for n, a, b in zip(names, base[1], fold_off[1]):
        mx = max(float(b.abs().max()), 1e-6)
        e_max, e_rms = float((a.double() - b.double()).abs().max()) / mx, float((a.double() - b.double()).pow(2).mean().sqrt()) / mx
        print(f"[fold on / off] {n}: max {e_max:.2e} rms {e_rms:.2e} of the maximum")
        if n in ("mu", "logvar"):
            assert e_max <= 5e-3, (n, e_max)
        else:
            assert e_max <= 6e-2 and e_rms <= 8e-3, (n, e_max, e_rms)


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_soak_every_iteration_a_new_batch(mode):
    """Real training changes the image contents and the object count O every iteration (the bench reuses one resident batch): ten
    iterations on ten different synthetic batches (11-16 images, O between ~50 and ~110) must keep every loss finite, keep the
    fused re-pack plan in use (no individual re-pack of a planned parameter), and must not grow the device memory after the first
    iterations have sized the workspaces (tools/soak.py is the long form of this test)."""
    import math
    from agl import lib as L, synth
    from agl.trainer import Trainer, batch_to_device
    nets = build_nets(False)
    tr = Trainer(*nets, torch.from_numpy(synth.make_pos_weight()), conv_dtype=mode)
    torch.cuda.reset_peak_memory_stats()
    mem, objs = [], []
    for i in range(10):
        bn = synth.make_batch(16 - (i % 6), 64, seed=500 + i)
        objs.append(int(bn["objs"].shape[0]))
        tr.step(batch_to_device(bn, DEV))
        tr.finish()
        torch.cuda.synchronize()
        d = tr.loss_dict()
        assert all(math.isfinite(v) for v in d.values()), (i, d)
        mem.append(torch.cuda.max_memory_allocated() / 2**30)
    assert len(set(objs)) >= 6, objs
    assert mem[-1] <= mem[6] * 1.10 + 0.25, f"device memory keeps growing over changing batches: {mem}"
    assert L.PACK_STATS["fused"] >= 20


def test_two_trainers_driven_from_two_host_threads():
    """the schedule of an iteration lives in module-level switches (BatchNorm tape, deferred updates, private
    gradient arenas, convolution flags, weight-gradient streams); Trainer.step() holds a process lock and restores them, so two
    trainers — different arithmetic modes, different batches — stepped from two host threads at once give exactly what each gives
    alone, and the switches are back to their defaults afterwards."""
    import threading
    from agl import functional as F
    from agl import lib as L
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    pw = torch.from_numpy(synth.make_pos_weight())
    cases = [("f32x3", synth.make_batch(3, 64, seed=5, objs_per_image=[3, 4, 2])), ("bf16", synth.make_batch(2, 64, seed=6, objs_per_image=[5, 3]))]

    def run(dtype, bn, out, steps=2):
        nets = build_nets(False)
        tr = Trainer(*nets, pw, conv_dtype=dtype)
        O = bn["objs"].shape[0]
        g = torch.Generator().manual_seed(17)
        eps = [torch.randn(O, 64, generator=g) for _ in range(6)]
        b = batch_to_device(bn, DEV)
        for _ in range(steps):
            tr.step(b, eps[:3], eps[3:])
        tr.finish()
        torch.cuda.synchronize()
        out.append((tr.loss_dict(), [p.detach().clone() for p in nets[0].parameters()]))

    alone = [[], []]
    for (dtype, bn), o in zip(cases, alone):
        run(dtype, bn, o)
    both = [[], []]
    ths = [threading.Thread(target=run, args=(dtype, bn, o)) for (dtype, bn), o in zip(cases, both)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for a, b in zip(alone, both):
        (la, pa), (lb, pb) = a[0], b[0]
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(la[k])), (k, la[k], lb[k])
        for x, y in zip(pa, pb):
            assert float((x - y).abs().max()) <= 2 * 2e-4 + 1e-7, "parameters differ by more than the Adam steps' reach"
    assert F.BN_TAPE is None and F.BN_DEFER is None and F.GRAD_ARENA is None and L.WGRAD_STREAMS is None and L.CONV_FLAGS == 0


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_fused_repack_after_the_optimiser_step_equals_individual_packs(mode):
    """agl.lib.PackPlan: after each Adam step the arena re-packs every packed form of its convolution weights in ONE launch
    (agl_conv2d_pack_many), in place, instead of one launch per weight and form at their first use in the next iteration.  After two
    iterations every cached pack must be current (no further individual pack of a parameter in the third iteration) and bit-equal to
    a fresh agl_conv2d_pack_weights of the weight as it is now."""
    from agl import lib as L, synth
    from agl.trainer import Trainer, batch_to_device
    nets = build_nets(False)
    tr = Trainer(*nets, torch.from_numpy(synth.make_pos_weight()), conv_dtype=mode)
    assert tr.flat_g.pack_plan is not None and tr.flat_d.pack_plan is not None
    b = batch_to_device(synth.make_batch(5, 64, seed=7), DEV)
    for _ in range(2):
        tr.step(b)
    tr.finish()
    torch.cuda.synchronize()
    n_g, n_d = len(tr.flat_g.pack_plan.entries), len(tr.flat_d.pack_plan.entries)
    assert n_g > 5 and n_d > 5 and L.PACK_STATS["fused"] >= 4, (n_g, n_d, L.PACK_STATS)      # (a small batch: many layers stay below the patch kernel's grid threshold)
    flags = {"f32x3": L.CONV_SPLIT3, "bf16": L.CONV_BF16}[mode]
    checked = 0
    for flat in (tr.flat_g, tr.flat_d):
        for (owner, key, wsrc, pass_, Cin, Cout, ks, stride, fl, buf, src) in flat.pack_plan.entries.values():
            hit = owner.__dict__["_agl_packs"][key]
            assert hit[1] is buf and hit[0] == wsrc.version(), "a pack of the plan is stale after the optimiser step"
            fresh = torch.empty_like(buf)
            L.call("agl_conv2d_pack_weights", L.ptr(src.detach()), fresh.data_ptr(), fresh.numel(), pass_, Cin, Cout, ks, stride, fl | L.CONV_ANY_GRID, L.stream())
            assert fl == flags
            assert torch.equal(fresh, buf), key
            checked += 1
    assert checked == n_g + n_d
    before = L.PACK_STATS["packs"]
    planned = {(id(e[0]), e[1]) for f in (tr.flat_g, tr.flat_d) for e in f.pack_plan.entries.values()}
    tr.step(b)
    tr.finish()
    torch.cuda.synchronize()
    planned_after = {(id(e[0]), e[1]) for f in (tr.flat_g, tr.flat_d) for e in f.pack_plan.entries.values()}
    assert planned_after == planned, "the third iteration packed a parameter individually again"
    assert L.PACK_STATS["packs"] - before < 40, L.PACK_STATS      # (derived weights only: ConvLSTM halves, pooled filters)


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_fused_repack_drops_the_packs_of_a_moved_parameter(mode):
    """PackPlan's descriptor table holds raw device pointers of the parameters and of the pack buffers.  When a parameter's
    storage moves after its pack was noted (a .data reassignment, module.to(), a re-flattened arena), the in-place re-pack would read
    the abandoned storage and stamp the entry as current — silently convolving with stale weights.  repack() must notice, drop the
    entry together with its cache entry, and the next use must pack the NEW storage on the ordinary miss path."""
    import torch.nn as nn
    from agl import lib as L
    from agl.flat import FlatParams
    flags = {"f32x3": L.CONV_SPLIT3, "bf16": L.CONV_BF16}[mode] | L.CONV_ANY_GRID
    torch.manual_seed(3)
    m = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(DEV)
    flat = FlatParams([m])
    assert flat.pack_plan is not None
    x = torch.randn(4, 64, 16, 16, device=DEV)
    w = m.weight
    with L.conv_flags(flags), torch.no_grad():
        y0 = L.conv2d_fwd(x, w, None, 1, 1, wsrc=w._agl_wsrc)
        assert len(flat.pack_plan.entries) == 1
        L.note_joined()
        flat.adam_step(1e-3, 0.5, 0.999, 1e-8)                 # (zero gradient: the weights stay; the plan re-packs in one launch)
        fused0, dropped0 = L.PACK_STATS["fused"], L.PACK_STATS["dropped"]
        y1 = L.conv2d_fwd(x, w, None, 1, 1, wsrc=w._agl_wsrc)
        assert torch.equal(y0, y1)
        new_vals = torch.randn_like(w) * 0.05
        w.data = new_vals.clone()                               # the parameter leaves the arena: its old storage keeps the old values
        L.note_joined()
        flat.pack_plan.repack()
        assert L.PACK_STATS["dropped"] == dropped0 + 1 and len(flat.pack_plan.entries) == 0
        assert not w.__dict__.get("_agl_packs"), "the stale cache entry survived"
        y2 = L.conv2d_fwd(x, w, None, 1, 1, wsrc=w._agl_wsrc)
    ref = TF.conv2d(x.cpu(), new_vals.cpu(), None, padding=1)
    tol = 2e-5 if mode == "f32x3" else 2e-2
    assert float((y2.cpu() - ref).abs().max()) <= tol * float(ref.abs().max()), "the convolution still ran with the abandoned storage's weights"


def test_concurrent_schedule_equals_sequential_schedule():
    """The training iteration with its concurrent schedule — discriminators on three streams, weight gradients on side streams,
    the generator's rand / shift branches on two streams with private gradient arenas and deferred BatchNorm updates, the G step's
    generator pass beside the D step — against the same iteration issued on ONE stream in program order (Trainer(streams=False)).
    Same kernels and arithmetic; only the order in which the generator's gradient arenas are added differs.  Two iterations each
    (the second runs on updated weights and re-packed weight caches).  Checked: the losses of both iterations (2e-5), every
    gradient tensor of the first iteration (relative L2 <= 1e-4; the crop backward's atomicAdd is the only unordered sum), the
    BatchNorm running statistics and spectral-norm vectors after the first iteration (1e-5; 5e-3 after the second, whose weights
    already differ by Adam's sign flips), and every parameter to within the two
    Adam steps' reach (|diff| <= 4 lr: an element whose gradient is rounding noise may step +-lr either way in each iteration)."""
    from agl import synth
    from agl.trainer import Trainer, batch_to_device, LR
    pw = torch.from_numpy(synth.make_pos_weight())
    bn = synth.make_batch(6, 64, seed=31)
    O = bn["objs"].shape[0]
    gen = torch.Generator().manual_seed(4)
    eps = [[torch.randn(O, 64, generator=gen) for _ in range(6)] for _ in range(2)]
    results = []
    for streams in (True, False):
        nets = build_nets(False)
        tr = Trainer(*nets, pw, conv_dtype="f32x3", streams=streams)
        assert (tr.d_streams is not None) == streams and (tr.g_streams is not None) == streams
        grads = {}

        def grab(tag, which):
            def f(t):
                if tag not in grads:
                    grads[tag] = [q.grad.detach().cpu().clone() for n in which for q in n.parameters()]
            return f

        tr.on_d_backward, tr.on_g_backward = grab("D", nets[1:]), grab("G", nets[:1])
        losses = []
        for it in range(2):
            tr.step(batch_to_device(bn, DEV), eps[it][:3], eps[it][3:])
            tr.finish()
            torch.cuda.synchronize()
            losses.append(tr.loss_dict())
            if it == 0:
                first = [{k: v.detach().cpu().clone() for k, v in n.state_dict().items()} for n in nets]
        state = [{k: v.detach().cpu().clone() for k, v in n.state_dict().items()} for n in nets]
        names = [[k for k, _ in n.named_parameters()] for n in nets]
        results.append((losses, state, grads, names, first))
    (la, sa, ga, names, fa), (lb, sb, gb, _, fb) = results
    for it in range(2):
        for k in la[it]:
            assert abs(la[it][k] - lb[it][k]) <= 2e-5 * max(1.0, abs(lb[it][k])), (it, k, la[it][k], lb[it][k])
    for tag in ("D", "G"):
        big = max(float(g.double().norm()) for g in gb[tag])
        for i, (a, b) in enumerate(zip(ga[tag], gb[tag])):
            nb = float(b.double().norm())
            if nb > 1e-4 * big:
                rel = float((a.double() - b.double()).norm()) / nb
                assert rel <= 1e-4, (tag, i, rel)
    for na, nb, pn in zip(sa, sb, names):
        for k in na:
            if not na[k].is_floating_point():
                assert torch.equal(na[k], nb[k]), k
            elif k in pn:
                assert float((na[k] - nb[k]).abs().max()) <= 4.0 * LR * 1.01, (k, float((na[k] - nb[k]).abs().max()))
            else:      # (second iteration: the weights already differ by Adam's sign flips)
                close(na[k], nb[k], 5e-3, "buffer after two iterations " + k)
    for na, nb, pn in zip(fa, fb, names):      # after the FIRST iteration the buffers saw identical weights: tight
        for k in na:
            if na[k].is_floating_point() and k not in pn:
                close(na[k], nb[k], 1e-5, "buffer after one iteration " + k)


def test_hinge_losses_vs_torch():
    """loss_hinge_dis / loss_hinge_gen (models/spade/networks/loss.py:65-76; off the reference's train path) against the
    torch-CPU arithmetic of GANLoss('hinge'): values and gradients <= 1e-6."""
    from models.discriminator import loss_hinge_dis, loss_hinge_gen
    g = torch.Generator().manual_seed(3)
    fake, real = torch.randn(37, generator=g) * 2, torch.randn(37, generator=g) * 2
    fr, rr = fake.clone().requires_grad_(True), real.clone().requires_grad_(True)
    ref = -torch.mean(torch.min(rr - 1, torch.zeros_like(rr))) - torch.mean(torch.min(-fr - 1, torch.zeros_like(fr)))
    ref.backward()
    fg, rg = fake.to(DEV).requires_grad_(True), real.to(DEV).requires_grad_(True)
    out = loss_hinge_dis(fg, rg)
    out.backward()
    assert abs(float(out) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    close(fg.grad, fr.grad, 1e-6, "d hinge / d fake")
    close(rg.grad, rr.grad, 1e-6, "d hinge / d real")
    f2 = fake.clone().requires_grad_(True)
    (-f2.mean()).backward()
    f3 = fake.to(DEV).requires_grad_(True)
    o3 = loss_hinge_gen(f3)
    o3.backward()
    assert abs(float(o3) + float(fake.mean())) <= 1e-6
    close(f3.grad, f2.grad, 1e-6, "d hinge_gen")
