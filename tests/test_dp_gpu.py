"""GPU: the data-parallel Trainer path end to end with two ranks sharing the one available GPU (gloo carries the
all-reduce, so this exercises the side-stream ordering, the rank-0 weight broadcast and the 1/world scaling exactly
as the RCCL run does).  With identical shards on both ranks the averaged gradients equal the single-rank ones, so the
two-rank run must reproduce the single-rank losses and weights; and both ranks must end with identical weights."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


def _setup():
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)


def _make(seed_offset):
    from agl import synth
    from agl.trainer import Trainer, batch_to_device
    from models.generator_obj_att import Generator
    from models.discriminator import ImageDiscriminator, ObjectDiscriminator, AttributeDiscriminator, add_sn
    torch.manual_seed(123 + seed_offset)        # ranks start from DIFFERENT weights: the broadcast must fix that
    nets = [Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=32, attribute_dim=106),
            add_sn(ImageDiscriminator(conv_dim=64)), add_sn(ObjectDiscriminator(n_class=179)), add_sn(AttributeDiscriminator(n_attribute=106))]
    nets = [m.to("cuda:0") for m in nets]
    tr = Trainer(*nets, torch.from_numpy(synth.make_pos_weight()))
    bn = synth.make_batch(2, 64, seed=21, objs_per_image=[3, 4])
    b = batch_to_device(bn, "cuda:0")
    g = torch.Generator().manual_seed(9)
    eps = [torch.randn(7, 64, generator=g) for _ in range(6)]
    return tr, b, eps


def _run(tr, b, eps, steps=2):
    for _ in range(steps):
        tr.step(b, eps[:3], eps[3:])
    tr.finish()
    torch.cuda.synchronize()
    return tr.loss_dict(), tr.flat_g.p.detach().clone(), tr.flat_d.p.detach().clone()


def _digest(t):
    """Small by-value summary of an arena: checksums + a strided sample."""
    t = t.detach().double().cpu()
    return float(t.sum()), float(t.abs().sum()), t[::997].clone().numpy()


def _worker(rank, world, port, q):
    _setup()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, b, eps = _make(seed_offset=rank)
    assert tr.sync.enabled and tr.sync.world == 2
    # with data parallelism step() returns while the G all-reduce + Adam still run on the side stream; a state_dict()
    # taken right after step() (what checkpoint.save_model does) must already hold the post-update weights
    tr.step(b, eps[:3], eps[3:])
    sd = {k: v.detach().clone() for k, v in tr.netG.state_dict().items()}          # no finish(), no synchronize()
    tr.finish()
    torch.cuda.synchronize()
    for k, v in tr.netG.state_dict().items():
        assert torch.equal(sd[k], v), ("state_dict taken right after step() is stale", k)
    # the all-reduce + Adam of an arena run on the side stream; the gradients other code may read (test probes,
    # gradient clipping a user adds) must be the LOCAL ones until that hand-off, and the main stream must not touch the weights
    # before the event step() leaves in _g_ready / _d_ready.  Probe: the local gradient arena seen in on_g_backward (main stream,
    # before the exchange) equals a copy taken there; after step() returns, _g_ready is set (work still pending on the side
    # stream) and finish() makes the main stream wait for it.
    tr, b, eps = _make(seed_offset=rank)
    seen = {}
    tr.on_g_backward = lambda t: seen.setdefault("g", t.flat_g.g.detach().clone())
    p_before = tr.flat_g.p.detach().clone()
    tr.step(b, eps[:3], eps[3:])
    assert tr._g_ready is not None, "step() must leave the G exchange pending on the side stream"
    tr.finish()
    assert tr._g_ready is None
    torch.cuda.synchronize()
    assert float(seen["g"].abs().sum()) > 0 and not torch.equal(tr.flat_g.p, p_before)
    summed = seen["g"].clone()
    dist.all_reduce(summed)                    # what the side stream exchanged: SUM over ranks of the local arenas
    assert torch.allclose(tr.flat_g.g, summed, rtol=0, atol=0), "the arena must hold exactly the all-reduced gradients after finish()"
    tr, b, eps = _make(seed_offset=rank)
    losses, pg, pd = _run(tr, b, eps)
    same = []
    for t in (pg, pd):                         # replicas must hold bit-identical weights
        ref = t.clone()
        dist.broadcast(ref, src=0)
        same.append(bool(torch.equal(ref, t)))
    q.put((rank, losses, _digest(pg), _digest(pd), same))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_rank():
    _setup()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    tr, b, eps = _make(seed_offset=0)           # rank 0's initial weights, single process
    l1, pg1, pd1 = _run(tr, b, eps)
    import numpy as np
    (_, l_r0, dg0, dd0, same0), (_, l_r1, dg1, dd1, same1) = res
    assert all(same0) and all(same1), "replicas diverged"
    for k, v in l1.items():
        assert abs(l_r0[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, l_r0[k], v)
        assert abs(l_r1[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, l_r1[k], v)
    # identical shards: (g + g) / 2 == g, so the two-rank weights equal the single-rank ones (up to the crop-scatter
    # atomics' run-to-run order and Adam sign flips of noise-level gradients: <= 2*lr per element)
    for (s0, a0, samp0), t1 in ((dg0, pg1), (dd0, pd1)):
        s1, a1, samp1 = _digest(t1)
        assert abs(a0 - a1) <= 1e-5 * a1
        assert np.abs(samp0 - samp1).max() <= 1e-3
