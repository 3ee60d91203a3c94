#!/usr/bin/env python
"""Benchmark of the G+D training step (BASELINE.json metric: images/sec per G+D train step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--res 64|128] [--dtype f32|bf16] [--batch B]

One process per GPU.  With `--gpus N` > 1 and no WORLD_SIZE in the environment this (still GPU-free) parent starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>` as a
child and relays its output; under torch.distributed.run (WORLD_SIZE set) it is a rank.  Each rank trains its own
shard (weak scaling: per-GPU batch fixed); the only exchange is the RCCL all-reduce of the two flat gradient arenas.
Rank 0 prints ONE JSON line.  The launch mode is decided before the first GPU call; nothing re-execs a process
that has touched the GPU.

Default workload = BASELINE config 2 (64 px, batch 64/GPU, fp32).  Without --res/--dtype the same line also carries
BASELINE config 3 (128 px, batch 32/GPU, bf16 MFMA) as `secondary`, so both halves of the metric are in one record.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "attribute-guided-image-generation-from-layout_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# algorithmic FLOPs (2*MAC, fwd + bwd as the reference graph executes them) per image per
# train step, linear in P = objects per image (FlopCounterMode on the reference step, exact fit at P=3,6,9).
FLOPS_PER_IMAGE = {64: (7.334e10, 6.566e10), 128: (6.202e11, 2.232e11)}
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: bf16 MFMA dense peak (~2.5 PF)
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
# Peak of the pipe a convolution launch ran on (agl_conv2d_last_pipe): 0 exact fp32 (fp32 MFMA / fp32 VALU, same 157.3 peak);
# 1 bf16 MFMA, one product per multiply-add; 3 = 16-bit MFMA with split operands: agl_conv2d_split_products() MFMA products per fp32
# multiply-add — THREE with the fp16 hi / lo form this library is built with (2500/3 = 833.3 fp32-equivalent TFLOP/s ceiling), six with
# the bf16 x3 form of rounds 2-4 (416.7).  set_split_form() fills entry 3 from the loaded library.
PIPE_PEAK = {0: PEAK_F32_MFMA_TFLOPS, 1: PEAK_BF16_MFMA_TFLOPS, 3: PEAK_BF16_MFMA_TFLOPS / 3.0}
PIPE_NAME = {0: "fp32 (MFMA 32x32x2 f32 / VALU)", 1: "bf16 MFMA", 3: "fp16 MFMA, hi/lo split operands (3 products per MAC)"}
# JSON dtype = how the multiply-adds are computed: "f32x3" = fp32 tensors and accumulation with every product formed from THREE 16-bit
# matrix-core products of two fp16 terms per operand (config.products spells it out first); "f32" = exact fp32 MFMA everywhere
DTYPE_NAME = {"f32": "f32", "f32x3": "f32x3", "bf16": "bf16"}
PRODUCTS = {
    "f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32) in every convolution",
    "f32x3": "fp16 hi/lo split products: fp32 tensors, 3 fp16-MFMA products per fp32 multiply-add under power-of-two block scales "
             "(fp32-accurate: error vs fp64 <= that of the exact fp32 MFMA kernels), fp32 accumulation",
    "bf16": "bf16 MFMA operands (rounded once when staged), fp32 accumulation; activations with convolution-only readers stored as bf16 "
            "(SPADE outputs in front of the decoder's layers, box-filtered maps, first-convolution outputs of the flat D blocks), the "
            "rest and all statistics / gradients fp32",
}
ARITHMETIC = {
    "f32": "fp32 tensors; every convolution on exact fp32 MFMA (v_mfma_f32_32x32x2_f32)",
    "f32x3": "fp32 tensors and fp32 accumulation; in the LDS-patch kernels (csrc/pconv.hip: 1x1/3x3/5x5 stride 1, 4x4/3x3 stride 2 "
             "forward, stride-1 and 4x4/stride-2 input gradients, weight gradients) each fp32 operand is carried as two fp16 terms "
             "(hi + lo' * 2^-11, ~23 bits) under a power-of-two scale per staged block (weights: per 16-channel chunk at pack time; "
             "activations: per staged chunk / tile from the registers being converted), hi*hi + hi*lo' + lo'*hi on the fp16 matrix "
             "cores, the hi*hi accumulator emptied into an fp32 total every ~32 instructions — measured as accurate as the fp32 MFMA "
             "chain (tests: error vs fp64 <= 2x that of the exact kernel, also at operand magnitudes 1e-8 and 1e4; whole step at this "
             "size vs the CPU oracle at the fp32 tolerances); exact fp32 MFMA in all other kernels",
    "bf16": "convolution operands rounded to bf16 when staged (or stored as bf16 by their producer when only convolutions read them: "
            "identical values), bf16 MFMA with fp32 accumulation; BatchNorm / ConditionalBatchNorm apply folded into the consuming "
            "convolution's staging pass; statistics, spectral norm, losses, gradients and Adam fp32",
}
CONV_NAMES = ("agl_conv2d_fwd", "agl_conv2d_fwd_stats", "agl_conv2d_bwd_data", "agl_conv2d_bwd_weight", "agl_conv2d_fwd_fold",
              "agl_conv2d_bwd_weight_fold", "agl_conv2d_fwd_addend", "agl_conv2d_fwd_shortcut")
NORM_NAMES = ("agl_bn_stats", "agl_bn_stats_from_partials", "agl_norm_apply_fwd", "agl_norm_bwd", "agl_norm_bwd_fold", "agl_norm_apply_fwd_y16",
              "agl_norm_bwd_y16")


def set_split_form():
    """Price pipe 3 (split operands) with the product count of the loaded library (agl_conv2d_split_products: 3 = fp16 hi/lo, 6 = bf16 x3)."""
    from agl import lib as _L
    sp = _L.load().agl_conv2d_split_products()
    PIPE_PEAK[3] = PEAK_BF16_MFMA_TFLOPS / sp
    if sp == 6:
        PIPE_NAME[3] = "bf16 MFMA, split operands (6 products per MAC)"
        PRODUCTS["f32x3"] = "bf16x3 split products: fp32 tensors, 6 bf16-MFMA products per fp32 multiply-add (fp32-accurate), fp32 accumulation"
    return sp


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=None, choices=[64, 128], help="default 64 (+ the 128 px bf16 secondary)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default 64 at 64px, 32 at 128px)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 128 px / bf16 secondary result")
    ap.add_argument("--dtype", default=None, choices=["f32", "f32x3", "bf16"],
                    help="arithmetic of the convolutions.  f32x3 (default for BASELINE config 2): fp32 tensors, fp32-accurate "
                         "products formed on the fp16 matrix cores from two fp16 terms per operand (three products, fp32 "
                         "accumulation) in the kernels of csrc/pconv.hip, exact fp32 MFMA elsewhere; f32: exact fp32 MFMA "
                         "everywhere; bf16: operands rounded to bf16 (configs 3/5).  Statistics, SN, losses, Adam: fp32 always")
    ap.add_argument("--seed", type=int, default=1234, help="synthetic batch seed (rank is added)")
    ap.add_argument("--vary-batch", type=int, default=4, metavar="K",
                    help="after the timed steps on the fixed batch, time the same number of steps cycling K pre-resident batches with "
                         "different object counts (real training changes O every batch: workspaces, pack plans and allocator state must "
                         "cope); reported as `vary_batch` beside the fixed-batch value; 0 = skip")
    ap.add_argument("--two-generator-passes", action="store_true",
                    help="evaluate the whole generator twice per iteration like the reference loop instead of reusing the "
                         "draw-independent parts of the first evaluation (identical results; reported for comparison)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: rendezvous, barrier, max-over-ranks timing and the JSON line, "
                         "with no training step (value is null); used by the CPU tests with --backend gloo")
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a) -> int:
    """GPU-free parent of a multi-rank run: one torch.distributed.run child (which starts the N ranks)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------- workloads
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
    return [m.to(dev) for m in (G, Di, Do, Da)], obj_size


def cpu_baseline(max_seconds=30.0):
    """The reference's arithmetic on the host cores: the oracle (plain PyTorch-CPU restatement of the reference graph,
    bit-exact against the imported reference in the build container) on BASELINE config 1 (64 px, batch 4)."""
    import torch
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
            "sample": f"oracle (PyTorch-CPU restatement of the reference graph) on BASELINE config 1: 64px batch 4, O={O}, "
                      f"{n} timed step(s) after 1 warm-up, {dt:.2f} s/step",
            "note": "config 1 is the reference's own CPU-runnable case (batch 4); the GPU value is config 2 (batch 64) — "
                    "compare per-image rates, the batches differ"}


def hbm_traffic(tag):
    """HBM bytes per iteration of the normalisation family and of the convolution family from the committed rocprofv3 PMC passes
    (FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md §HBM prescribes; tools/hbm_table.py writes the file).  This round's
    file; None when it was not collected for the loaded library's ABI / split form."""
    try:
        with open(os.path.join(ROOT, "profiles", "r05_hbm_traffic.json")) as f:
            d = json.load(f).get(tag)
    except (OSError, ValueError):
        return None
    if not d:
        return None
    # a counter profile describes the kernels it was collected on: quote it only for a library of the same ABI and split form
    # (the figure must not go stale silently); `collected_at` names the commit
    from agl import lib as _L
    lib = _L.load()
    if d.get("abi") != lib.agl_version() or d.get("split_products") != lib.agl_conv2d_split_products():
        return None
    return d


def run_workload(a, res, dtype, per_gpu, steps, warmup, dev, rank, world, dist):
    """Time `steps` training iterations of one workload; returns the result fields of the JSON line (rank 0) or None."""
    import numpy as np
    import torch
    from agl import lib as L, synth
    from agl.trainer import Trainer, batch_to_device
    torch.manual_seed(0)                         # identical initial weights on every rank
    nets, obj_size = build_nets(res, dev)
    pw = torch.from_numpy(synth.make_pos_weight())
    # attribute_est is derived on device from the pre-step D_att logits, as the reference loop does (train64.py:156-166)
    tr = Trainer(*nets, pw, estimate_attributes=True, reuse_generator_pass=not a.two_generator_passes, conv_dtype=dtype)
    # one rank: the plain draw (P ~ U{3..9} per image); N ranks: the global batch's images dealt so that the ranks' object counts match
    # (agl.synth.balanced_object_counts) — the weak-scaling step is as slow as the rank with the most objects
    counts = synth.balanced_object_counts(per_gpu, world, seed=a.seed)[rank] if world > 1 else None
    bn = synth.make_batch(per_gpu, res, seed=a.seed + rank, objs_per_image=counts)
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

    for _ in range(warmup):
        one_step()
    tr.finish()
    fence()
    calls0 = L.CALL_COUNT
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    tr.finish()
    t_host = time.perf_counter() - t0            # host time to enqueue the K steps (the GPU may still be running)
    fence()
    dt = time.perf_counter() - t0
    abi_calls = (L.CALL_COUNT - calls0) / max(1, steps)
    objs = [O]
    vary = None
    if a.vary_batch and a.vary_batch > 1:
        # K batches with different image contents and object counts, all resident before the clock starts (no H2D in the timed region)
        K = a.vary_batch
        pool = []
        for i in range(K):
            bi = synth.make_batch(per_gpu, res, seed=a.seed + rank + 1000 * (i + 1))
            Oi = int(bi["objs"].shape[0])
            gi = torch.Generator().manual_seed(200 + rank + i)
            pool.append((batch_to_device(bi, dev), [torch.randn(Oi, 64, generator=gi).to(dev) for _ in range(6)], Oi))
        for bb, ee, _ in pool:                     # one untimed pass: every object count has been seen once (allocations, plans)
            tr.step(bb, ee[:3], ee[3:])
        tr.finish()
        fence()
        tv = time.perf_counter()
        for i in range(steps):
            bb, ee, _ = pool[i % K]
            tr.step(bb, ee[:3], ee[3:])
        tr.finish()
        fence()
        dv = time.perf_counter() - tv
        if world > 1:
            t = torch.tensor([dv], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dv = float(t.item())
        vary = {"k": K, "value": round(per_gpu * world * steps / dv, 3), "unit": "images/s", "ms_per_step": round(1e3 * dv / steps, 3),
                "objects_per_batch_rank0": [o for _, _, o in pool], "peak_device_memory_gib": round(torch.cuda.max_memory_allocated() / 2**30, 2)}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ot = torch.zeros(world, dtype=torch.int64, device=dev)
        ot[rank] = O
        dist.all_reduce(ot)
        objs = [int(v) for v in ot.tolist()]
    losses = tr.loss_dict()
    assert all(np.isfinite(v) for v in losses.values()), losses

    roof = roof_hbm = None
    if not a.no_roofline:
        # one extra, instrumented step (outside the timed region; EVERY rank runs it, the gradient all-reduce is
        # collective): HIP events around every convolution launch (>99.9 % of the FLOPs) and every normalisation-family
        # launch (the dominant HBM-bound kernels) on the stream they are launched on; rank 0 reports.
        # The timed steps run the discriminator / generator-branch / weight-gradient chains on several streams; here every launch
        # is timed on its own — one stream, program order (Trainer.serial) — so that a launch's duration is the kernel's, not the
        # kernel's share of a GPU it divides with two other chains.
        with tr.serial():
            # The instrumented step must run with the GPU BEHIND the host: every kernel already queued when its start event
            # executes, so that an event pair brackets the kernel(s) of its call and not the host's latency between recording the
            # event and launching (with the GPU waiting on the host, the same 1365 launches read 104 ms or 136 ms depending on the
            # box's host).  One ordinary step (the one-stream schedule's allocations), then a spin kernel of about 2.5 step times
            # (torch.cuda._sleep, calibrated here) holds the stream while the host enqueues the whole instrumented step behind it.
            one_step()
            tr.finish()
            fence()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(); torch.cuda._sleep(2_000_000); c1.record(); torch.cuda.synchronize()
            cycles_per_ms = 2_000_000 / max(1e-3, c0.elapsed_time(c1))
            torch.cuda._sleep(int(min(800.0, 2.5 * 1e3 * dt / steps) * cycles_per_ms))
            L.EVENT_LOG = [] if rank == 0 else None
            packs0, fused0 = L.PACK_STATS["packs"], L.PACK_STATS["fused"]
            one_step()
            tr.finish()
            fence()
            log, L.EVENT_LOG = L.EVENT_LOG, None
            packs_step, fused_step = L.PACK_STATS["packs"] - packs0, L.PACK_STATS["fused"] - fused0      # (of the instrumented step alone)
            # the same one-stream schedule without the per-launch events: its step time over the timed (concurrent) schedule's is the
            # overlap the chains on several streams buy — what a kernel-stats profile, which serialises them, cannot show
            t1 = time.perf_counter()
            for _ in range(3):
                one_step()
            tr.finish()
            fence()
            serial_ms = 1e3 * (time.perf_counter() - t1) / 3
    if not a.no_roofline and rank == 0:
        set_split_form()
        conv = [e for e in log if e[0] in CONV_NAMES]
        conv_ms = sum(e[1].elapsed_time(e[2]) for e in conv)
        executed = sum(e[3] for e in conv)
        if os.environ.get("AGL_DUMP_CONV"):
            import collections
            agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
            only = os.environ.get("AGL_DUMP_PIPE")             # e.g. 0: the launches that stayed on the exact-fp32 kernels
            for name, e0, e1, f, dims, _pipe, _bytes in conv:
                if only is not None and str(_pipe) != only:
                    continue
                k = (name.replace('agl_conv2d_', ''), dims)
                agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1); agg[k][2] += f
            by_ms = os.environ.get("AGL_DUMP_SORT") == "ms"
            ref = float(os.environ.get("AGL_DUMP_REF_TF", "157.3")) * 1e9      # "lost" = time above what the launch would take at this rate
            rows = sorted(agg.items(), key=lambda kv: -(kv[1][1] if by_ms else kv[1][1] - kv[1][2] / ref))
            for (nm, dims), (cnt, ms, fl) in rows[:int(os.environ.get("AGL_DUMP_CONV_ROWS", "40"))]:
                print(f'{nm:11s} x{cnt:3d} {ms:7.2f} ms  {fl/ms/1e9 if ms else 0:6.1f} TF  lost {ms - fl/ref:6.2f} ms  dims {dims}', file=sys.stderr)
        c0, c1 = FLOPS_PER_IMAGE[res]
        flops_step = per_gpu * c0 + O * c1                      # algorithmic (reference graph), per GPU per step
        # Utilisation of the pipes actually used: every convolution launch is priced against the peak of the pipe its main kernel
        # ran on (the C ABI reports it per call: agl_conv2d_last_pipe), frac = sum_launches(executed FLOPs / peak of its pipe) /
        # sum_launches(measured time) = time the launches would take at peak / time they took: <= 1 by construction.  Executed
        # FLOPs = C-ABI agl_conv2d_*_flops (dense 2*MAC count minus the padded taps the position-major path skips).  `achieved`
        # is quoted on the bf16 matrix pipe (frac x 2500), where > 80 % of the time is spent in both arithmetic modes; by_pipe has
        # the per-pipe numbers.  The reference graph's FLOPs over the same time are reported separately (algorithmic_equiv_tflops):
        # the legal savings of DESIGN.md §3 make that figure larger than what the hardware executes: throughput, not utilisation.
        by_pipe, t_at_peak = {}, 0.0
        for pipe in sorted({e[5] for e in conv}):
            sel = [e for e in conv if e[5] == pipe]
            ms = sum(e[1].elapsed_time(e[2]) for e in sel)
            fl = sum(e[3] for e in sel)
            t_at_peak += fl / (PIPE_PEAK[pipe] * 1e12)
            by_pipe[PIPE_NAME[pipe]] = {"launches": len(sel), "ms": round(ms, 3), "executed_tflops": round(fl / (ms * 1e-3) / 1e12, 2) if ms else None,
                                        "peak_tflops": round(PIPE_PEAK[pipe], 1), "frac": round(fl / (PIPE_PEAK[pipe] * 1e12) / (ms * 1e-3), 4) if ms else None}
        frac = t_at_peak / (conv_ms * 1e-3)
        # What binds each launch: its algorithmic HBM bytes — every operand once at its stored width (agl.lib.bytes_of) —
        # give a second lower bound, bytes / 8 TB/s, beside FLOPs / pipe peak; the launch's roof is the larger of the two.
        # frac_binding = sum(roof time) / sum(measured time); hbm_roofed_share = share of the family's measured time spent in launches
        # whose byte bound exceeds their matrix-pipe bound (below the ridge point of their pipe).
        t_roof = ms_hbm_roofed = alg_bytes = 0.0
        for e in conv:
            t_m, t_h = e[3] / (PIPE_PEAK[e[5]] * 1e12), e[6] / (PEAK_HBM_GBS * 1e9)
            t_roof += max(t_m, t_h)
            alg_bytes += e[6]
            if t_h > t_m:
                ms_hbm_roofed += e[1].elapsed_time(e[2])
        # achieved = executed (fp32-equivalent) FLOPs over the launches' time — comparable from round to round and between arithmetic
        # modes; peak = the FLOP-weighted harmonic mean of the peaks of the pipes the launches ran on, so that
        # frac = achieved / peak = time at peak / time taken.  The bf16 issue rate (split products counted six times) is its own key.
        achieved = executed / (conv_ms * 1e-3) / 1e12
        main_pipe = max(by_pipe.values(), key=lambda v: v["ms"])
        on_bf16 = main_pipe["peak_tflops"] != round(PEAK_F32_MFMA_TFLOPS, 1)
        tr_conv = hbm_traffic(f"{res}_{dtype}")
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(achieved / frac, 1), "unit": "TFLOP/s",
                "frac": round(frac, 4),
                "frac_binding": round(t_roof / (conv_ms * 1e-3), 4),
                "hbm_roofed_share_of_time": round(ms_hbm_roofed / conv_ms, 4) if conv_ms else None,
                "algorithmic_bytes_per_launch": round(alg_bytes / max(1, len(conv))),
                "traffic": round(tr_conv["conv_bytes_per_iteration"] / max(1, len(conv))) if tr_conv and "conv_bytes_per_iteration" in tr_conv else None,
                "traffic_source": (f'{tr_conv.get("source")} @ {tr_conv.get("collected_at")}' if tr_conv and "conv_bytes_per_iteration" in tr_conv else None),
                "achieved_bf16_issue": round(frac * PEAK_BF16_MFMA_TFLOPS, 2) if on_bf16 else None,
                "definition": "achieved = executed fp32-equivalent FLOPs of all convolution launches / their measured time (one stream, HIP events per "
                              "launch); peak = FLOP-weighted harmonic mean of the peak of the pipe each launch ran on (fp32 157.3, bf16 MFMA 2500, "
                              f"split-operand 16-bit MFMA 2500/{int(round(PEAK_BF16_MFMA_TFLOPS / PIPE_PEAK[3]))} fp32-equivalent); frac = achieved / peak = time at peak / time taken; "
                              f"achieved_bf16_issue = frac x 2500 (16-bit MFMA issue rate: split products count {int(round(PEAK_BF16_MFMA_TFLOPS / PIPE_PEAK[3]))}x); traffic = HBM bytes per "
                              "launch of the family from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE); algorithmic_bytes_per_launch = "
                              "every operand of a launch once at its stored width, family average (traffic / that = wasted-traffic ratio); "
                              "frac_binding = sum over launches of max(FLOPs / pipe peak, algorithmic bytes / 8 TB/s) / measured time: the "
                              "utilisation of whichever roof binds each launch; hbm_roofed_share_of_time = share of the measured time in "
                              "launches whose byte bound is the larger one",
                "by_pipe": by_pipe,
                "kernel": "convolution family: pconv_k / pbww_k (bf16 matrix cores) + igemm_f32<Fwd|BwdData|BwdWeight|Pos*> + patch_conv + "
                          "few_bww_k / small_cout_conv incl. weight packing and slab / split-K reductions (all agl_conv2d_* launches of one step)",
                "launches_per_step": len(conv), "kernel_ms_per_step": round(conv_ms, 3),
                "executed_flops_per_step": executed, "algorithmic_flops_per_step": flops_step,
                "algorithmic_equiv_tflops": round(flops_step / (conv_ms * 1e-3) / 1e12, 3),
                "weight_packs_per_step": packs_step,      # individual pack launches (derived weights: ConvLSTM halves, pooled filters)
                "fused_repacks_per_step": fused_step}     # launches of agl.lib.PackPlan.repack in the instrumented step (one per arena)
        nrm = [e for e in log if e[0] in NORM_NAMES]
        nrm_ms = sum(e[1].elapsed_time(e[2]) for e in nrm)
        nbytes = sum(e[3] for e in nrm)
        if nrm_ms > 0:
            gbs = nbytes / (nrm_ms * 1e-3) / 1e9
            tag = f"{res}_{dtype}"
            tr_ = hbm_traffic(tag)
            roof_hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4),
                        "traffic": round(tr_["bytes_per_iteration"] / max(1, len(nrm))) if tr_ else None,
                        "traffic_source": (f'{tr_["source"]} @ {tr_.get("collected_at")}' if tr_ else None),
                        "algorithmic_bytes_per_launch": round(nbytes / max(1, len(nrm))),
                        "kernel": "normalisation family: bn_stats_partial/final + norm_apply_fwd + norm_bwd_rows/channels/apply "
                                  "(agl_bn_stats, agl_norm_apply_fwd, agl_norm_bwd launches of one step; algorithmic bytes of "
                                  "SURVEY 8d per call / event time per call)",
                        "launches_per_step": len(nrm), "kernel_ms_per_step": round(nrm_ms, 3)}
    del tr, nets, b, eps
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    if roof is not None:
        roof["serial_ms_per_step"] = round(serial_ms, 3)      # the same kernels on one stream in program order
        roof["overlap_factor"] = round(serial_ms / (1e3 * dt / steps), 3)
    images = per_gpu * world * steps
    return {"value": round(images / dt, 3), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup,
            "dtype": DTYPE_NAME[dtype],
            "config": {"products": PRODUCTS[dtype], "arithmetic": ARITHMETIC[dtype], "workload": f"{res}x{res} G+D train step, batch={per_gpu}/GPU, {DTYPE_NAME[dtype]}, synthetic VG-shaped batch "
                                   f"(P~U{{3..9}}), random-init weights; the batch and the pinned eps draws are resident in HBM "
                                   f"and reused every step (no H2D in the timed region)",
                       "global_batch": per_gpu * world, "objects_per_rank": objs, "parallelism": f"dp{world}",
                       "generator_schedule": "two full passes" if a.two_generator_passes else "draw-independent parts evaluated once",
                       "streams": "3 discriminator chains + 2-3 generator branches + weight-gradient side streams (one HIP stream each; "
                                  "AGL_D_STREAMS / AGL_G_STREAMS / AGL_WGRAD_STREAM=0 for the single-stream schedule); roofline launches "
                                  "are timed on one stream",
                       "abi_calls_per_step": round(abi_calls),
                       # host time to enqueue the K steps; the HIP queue throttles the host to the GPU's pace, so this is an upper
                       # bound of the host cost (measured un-throttled at batch 2: 60 ms per iteration, tools/host_profile.py)
                       "host_enqueue_ms_per_step_upper_bound": round(1e3 * t_host / steps, 3)},
            "roofline": roof, "roofline_hbm": roof_hbm, "vary_batch": vary}


def main():
    a = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(launch_ranks(a))                 # decided before any GPU call; the parent never touches the GPU
    world = int(env_world or "1")
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; run `python bench.py --gpus {a.gpus}` (self-launching) or "
              f"torch.distributed.run --nproc-per-node {a.gpus}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if a.dry_run:
        from agl import synth
        if world > 1:
            dist.init_process_group(a.backend, rank=rank, world_size=world)
            dist.barrier()
        objs_rank = [int(c.sum()) for c in synth.balanced_object_counts(64, world, seed=a.seed)] if world > 1 else None
        t0 = time.perf_counter()
        time.sleep(0.01 * (rank + 1))
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"metric": "images/sec per G+D train step (64px)", "value": None, "unit": "images/s", "n_gpus": world,
                              "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f32", "data": "none (launcher dry run, no training step)",
                              "dry_run": True, "max_rank_seconds": float(t.item()),
                              "config": {"workload": "dry run", "parallelism": f"dp{world}", "objects_per_rank": objs_rank}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))   # one GPU per rank on a real node
    torch.cuda.set_device(dev)
    if world > 1:
        if a.backend == "nccl":
            # RCCL's own stream at high priority: its kernels then do not queue behind the compute chains of the step's other streams
            # (the arena exchange sits between the D step's backward and the G step's discriminator passes: its latency is exposed)
            kw = {}
            try:
                from torch.distributed import ProcessGroupNCCL
                kw["pg_options"] = ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                kw = {}
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, **kw)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    from agl import lib as L
    L.load()
    set_split_form()
    res = a.res or 64
    dtype = a.dtype or "f32x3"
    per_gpu = a.batch or (64 if res == 64 else 32)
    main_r = run_workload(a, res, dtype, per_gpu, a.steps, a.warmup, dev, rank, world, dist)
    second = exact = None
    if a.res is None and a.dtype is None and a.batch is None and not a.no_secondary:
        # BASELINE config 3 (config 5 when N > 1): the 128 px half of the metric, bf16 MFMA convolutions
        second = run_workload(a, 128, "bf16", 32, min(a.steps, 10), min(a.warmup, 3), dev, rank, world, dist)
        # the headline workload once more with exact fp32 MFMA in every convolution (v_mfma_f32_32x32x2_f32), for comparison
        a_fixed = argparse.Namespace(**vars(a))
        a_fixed.vary_batch = 0
        exact = run_workload(a_fixed, 64, "f32", 64, min(a.steps, 6), min(a.warmup, 2), dev, rank, world, dist)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        out = {"metric": f"images/sec per G+D train step ({res}px)", "value": main_r["value"], "unit": "images/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": main_r["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[dtype], "data": "synthetic",
               "config": main_r["config"], "roofline": main_r["roofline"], "roofline_hbm": main_r["roofline_hbm"],
               "cpu_baseline": cpu, "vary_batch": main_r.get("vary_batch")}
        if second is not None:
            second["metric"] = "images/sec per G+D train step (128px)"
            second["unit"] = "images/s"
            exact["metric"] = "images/sec per G+D train step (64px), exact fp32 MFMA in every convolution"
            exact["unit"] = "images/s"
            out["secondary"] = [second, exact]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
