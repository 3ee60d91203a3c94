"""CPU, world_size 2 over gloo: the data-parallel exchange (agl.dp.GradSync) — SUM all-reduce of a flat
gradient arena, 1/world scaling, rank-0 broadcast of weights, and image sharding of a global batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from agl import synth
    from agl.dp import GradSync
    sync = GradSync()
    assert sync.enabled and sync.world == world and abs(sync.grad_scale - 1.0 / world) < 1e-12
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    sync.all_reduce_(g)
    w = torch.full((10,), float(rank))
    sync.broadcast_(w, 0)
    full = synth.make_batch(4, 64, seed=9)
    sh = synth.shard(full, rank, world)
    n_obj = torch.tensor([float(sh["objs"].shape[0])])
    dist.all_reduce(n_obj)
    out.put((rank, float(g[7]), float(w[3]), float(n_obj), int(full["objs"].shape[0]), sh["imgs"].shape[0]))
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, g7, w3, nobj, total, nimg in res:
        assert g7 == 7.0 * (1 + 2)          # SUM over ranks
        assert w3 == 0.0                    # rank 0's weights everywhere
        assert nobj == total and nimg == 2  # shards partition the objects; 2 images each


def test_gradsync_disabled_without_process_group():
    from agl.dp import GradSync
    s = GradSync()
    assert not s.enabled and s.world == 1 and s.grad_scale == 1.0
    t = torch.ones(4)
    assert s.all_reduce_(t) is t
