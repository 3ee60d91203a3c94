#!/usr/bin/env python
"""Benchmark of the G+D training step (BASELINE.json metric: images/sec per G+D train step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--res 64|128] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; with N>1 each rank trains its own shard (weak scaling: per-GPU batch fixed) and the
only exchange is the RCCL all-reduce of the two flat gradient arenas.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic FLOPs (2*MAC, fwd + bwd as the reference graph executes them) per image per
# train step, linear in P = objects per image (FlopCounterMode on the reference step, exact fit at P=3,6,9).
FLOPS_PER_IMAGE = {64: (7.334e10, 6.566e10), 128: (6.202e11, 2.232e11)}
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: bf16 MFMA dense peak (~2.5 PF)


def build_nets(res, dev):
    from models.discriminator import ImageDiscriminator, ObjectDiscriminator, AttributeDiscriminator, AttributeDiscriminator128, add_sn
    if res == 128:
        from models.generator_obj_att128 import Generator
        att, obj_size = AttributeDiscriminator128, 64
    else:
        from models.generator_obj_att import Generator
        att, obj_size = AttributeDiscriminator, 32
    G = Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=obj_size, attribute_dim=106)
    Di, Do, Da = add_sn(ImageDiscriminator(conv_dim=64)), add_sn(ObjectDiscriminator(n_class=179)), add_sn(att(n_attribute=106))
    cpu_state = None
    return [m.to(dev) for m in (G, Di, Do, Da)], obj_size


def cpu_baseline(max_seconds=45.0):
    """The reference's arithmetic on the host cores: the oracle (plain PyTorch-CPU restatement of the reference graph,
    bit-exact against the imported reference in the build container) on BASELINE config 1 (64 px, batch 4)."""
    from agl import synth
    from models.generator_obj_att import Generator
    from models.discriminator import ImageDiscriminator, ObjectDiscriminator, AttributeDiscriminator, add_sn
    import oracle.step as OS
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, 32))          # batch-4 layers stop scaling beyond ~32 threads (128 were slower than 8)
    torch.set_num_threads(threads)
    nets = [Generator(num_embeddings=179, obj_att_dim=64, z_dim=64, clstm_layers=3, obj_size=32, attribute_dim=106),
            add_sn(ImageDiscriminator(conv_dim=64)), add_sn(ObjectDiscriminator(n_class=179)), add_sn(AttributeDiscriminator(n_attribute=106))]
    ob = OS.OracleBackend(*[m.state_dict() for m in nets], res128=False, obj_size=32)
    bn = synth.make_batch(4, 64, seed=1234)
    b = {k: torch.from_numpy(v) for k, v in bn.items()}
    pw = torch.from_numpy(synth.make_pos_weight())
    O = bn["objs"].shape[0]
    eps = [torch.randn(O, 64) for _ in range(3)]
    t0 = time.time()
    OS.run_step(ob, b, pw, eps, eps)            # warm-up
    warm = time.time() - t0
    n, t1 = 0, time.time()
    while n < 1 or (time.time() - t1 + warm) < max_seconds and n < 4:
        OS.run_step(ob, b, pw, eps, eps)
        n += 1
    dt = (time.time() - t1) / n
    return {"value": round(4.0 / dt, 4), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"oracle (PyTorch-CPU restatement of the reference graph) 64px batch 4, O={O}, {n} timed step(s) after 1 warm-up, {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=64, choices=[64, 128])
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 64 at 64px, 32 at 128px)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="MFMA arithmetic of the convolutions: exact fp32 (BASELINE config 2) or bf16 operands with fp32 "
                         "accumulation (configs 3/5); statistics, SN, losses and Adam are fp32 either way")
    ap.add_argument("--seed", type=int, default=1234, help="synthetic batch seed (rank is added)")
    ap.add_argument("--two-generator-passes", action="store_true",
                    help="evaluate the whole generator twice per iteration like the reference loop instead of reusing the "
                         "draw-independent parts of the first evaluation (identical results; reported for comparison)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))   # one GPU per rank on a real node
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"

    from agl import lib as L, synth
    from agl.trainer import Trainer, batch_to_device
    L.load()
    L.set_conv_precision(a.dtype)
    per_gpu = a.batch or (64 if a.res == 64 else 32)
    torch.manual_seed(0)                         # identical initial weights on every rank
    nets, obj_size = build_nets(a.res, dev)
    pw = torch.from_numpy(synth.make_pos_weight())
    # attribute_est is derived on device from the pre-step D_att logits, as the reference loop does (train64.py:156-166)
    tr = Trainer(*nets, pw, estimate_attributes=True, reuse_generator_pass=not a.two_generator_passes)
    bn = synth.make_batch(per_gpu, a.res, seed=a.seed + rank)
    b = batch_to_device(bn, dev)
    O = int(bn["objs"].shape[0])
    gen = torch.Generator().manual_seed(100 + rank)
    eps = [torch.randn(O, 64, generator=gen).to(dev) for _ in range(6)]     # pinned draws, resident on the device

    def one_step():
        tr.step(b, eps[:3], eps[3:])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    tr.finish()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    tr.finish()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = tr.loss_dict()
    assert all(np.isfinite(v) for v in losses.values()), losses

    roof = None
    if not a.no_roofline:
        # one extra, instrumented step (outside the timed region; EVERY rank runs it, the gradient all-reduce is
        # collective): HIP events around every convolution launch (igemm_f32 family, >99.9 % of the algorithmic
        # FLOPs) on the stream they are launched on; rank 0 reports.
        L.EVENT_LOG = [] if rank == 0 else None
        one_step()
        tr.finish()
        fence()
        log, L.EVENT_LOG = L.EVENT_LOG, None
    if not a.no_roofline and rank == 0:
        conv_ms = sum(e[1].elapsed_time(e[2]) for e in log)
        executed = sum(e[3] for e in log)
        if os.environ.get("AGL_DUMP_CONV"):
            import collections
            agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
            for name, e0, e1, f, dims in log:
                k = (name.replace('agl_conv2d_', ''), dims)
                agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1); agg[k][2] += f
            rows = sorted(agg.items(), key=lambda kv: -(kv[1][1] - kv[1][2] / 157.3e9))
            for (nm, dims), (cnt, ms, fl) in rows[:int(os.environ.get("AGL_DUMP_CONV_ROWS", "40"))]:
                print(f'{nm:11s} x{cnt:3d} {ms:7.2f} ms  {fl/ms/1e9 if ms else 0:6.1f} TF  lost {ms - fl/157.3e9:6.2f} ms  dims {dims}', file=sys.stderr)
        c0, c1 = FLOPS_PER_IMAGE[a.res]
        flops_step = per_gpu * c0 + O * c1                      # algorithmic, per GPU per step
        ach = flops_step / (conv_ms * 1e-3) / 1e12
        peak = PEAK_F32_MFMA_TFLOPS if a.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": None,
                "kernel": "convolution family: igemm_f32<Fwd|BwdData|BwdWeight> + patch_conv + small_cout_conv (all agl_conv2d_* launches of one step)",
                "launches_per_step": len(log), "kernel_ms_per_step": round(conv_ms, 3),
                "executed_flops_per_step": executed, "executed_tflops": round(executed / (conv_ms * 1e-3) / 1e12, 3),
                "algorithmic_flops_per_step": flops_step}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        images = per_gpu * world * a.steps
        out = {"metric": f"images/sec per G+D train step ({a.res}px)", "value": round(images / dt, 3), "unit": "images/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{a.res}x{a.res} G+D train step, batch={per_gpu}/GPU, {a.dtype}, synthetic VG-shaped batch "
                                      f"(P~U{{3..9}}, O={O} objects on rank 0), random-init weights",
                          "global_batch": per_gpu * world, "objects_rank0": O, "parallelism": f"dp{world}", "generator_schedule": "two full passes" if a.two_generator_passes else "draw-independent parts evaluated once"},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
