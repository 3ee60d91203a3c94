"""One G+D training iteration on the HIP path — the harness counterpart of the reference's training
loop body (train64.py:160-161 pre-step D_att forward, :191-262 D step, :280-370 G step; train128.py
is the same with the 128 px modules).  See SURVEY.md §8(H).

What is kept exactly: the order of network calls (so BatchNorm running statistics and spectral-norm
u/v advance identically), the loss set and weights, Adam(2e-4, (.5,.999)) on all four networks.
Legal savings taken: the generator is evaluated twice per iteration with unchanged weights (:195 and :280) and only
the reconstruction branch depends on the fresh random draws, so the draw-independent parts (crop-encoder trunks,
attribute encoder, the `rand` and `shift` branches) are evaluated ONCE, their graph is kept across the D step, and
the second evaluation recomputes the reconstruction branch only; the BatchNorm running statistics of the reused
layers are advanced a second time by replaying their statistics kernels (agl.functional.BN_TAPE), in the
reference's per-layer order.  The D-step reconstruction branch builds no autograd graph; D weight gradients are not computed in the G step (the
reference zeroes them before use, :254-256); loss values/gradients come from fused kernels; the three
D optimisers are one fused Adam launch over one flat arena (identical hyper-parameters and step count).
Host data preparation (attribute estimate loop :162-166, attribute swap :170-188) is the batch
builder's job (agl.synth); the batch carries attribute / attribute_gt / attribute_est.
"""
from __future__ import annotations

import math
import threading
import time
import os
from typing import Dict, Optional, Sequence

import torch

G_REC_STREAM = False      # the G step's reconstruction branch on a stream of its own (closed experiment, round 3)

from . import functional as F
from . import lib as L
from . import losses as LS
from .dp import GradSync
from .flat import FlatParams

LAMBDAS = dict(img_adv=1.0, obj_adv=1.0, obj_cls=1.0, z_rec=8.0, img_rec=1.0, kl=0.01, att_cls=2.0)   # train64.py:439-446
LR, BETA1, BETA2, ADAM_EPS = 2e-4, 0.5, 0.999, 1e-8                                                   # train64.py:434,111-114
MIX = (0.4, 0.4, 0.2)

RAW = ["d_img_rec", "d_img_rand", "d_img_shift", "d_img_real", "d_obj_rec", "d_obj_rand", "d_obj_shift", "d_obj_real",
       "d_obj_cls", "d_att", "g_img_rec", "g_z_rand", "g_z_shift", "g_kl", "g_img_adv_rec", "g_img_adv_rand",
       "g_img_adv_shift", "g_obj_adv_rec", "g_obj_adv_rand", "g_obj_adv_shift", "g_obj_cls_rec", "g_obj_cls_rand",
       "g_obj_cls_shift", "g_att_rec", "g_att_rand", "g_att_shift"]
IDX = {k: i for i, k in enumerate(RAW)}

BATCH_KEYS = ("imgs", "objs", "boxes", "masks", "z", "attribute", "masks_shift", "boxes_shift", "attribute_est",
              "attribute_gt")


def batch_to_device(batch_np: Dict, device) -> Dict[str, torch.Tensor]:
    """numpy batch (agl.synth.make_batch) -> device tensors; obj_to_img stays on the CPU like the reference
    (train64.py:149-151 does not move it)."""
    out = {k: torch.from_numpy(batch_np[k]).to(device) for k in BATCH_KEYS}
    out["obj_to_img"] = torch.from_numpy(batch_np["obj_to_img"])
    return out


# One iteration at a time per process.  The schedule of an iteration lives in module-level switches of agl.functional / agl.lib
# (BatchNorm tape, deferred updates, private gradient arenas, convolution flags, weight-gradient stream map) that its forward passes
# set and its backward passes — run by autograd's device thread — read; they are set and restored inside step(), so trainers that
# alternate on one thread never see each other's state, and this lock makes two trainers driven from two host threads take turns
# instead of interleaving (the design is one process per GPU with one host thread issuing).
_STEP_LOCK = threading.RLock()


class Trainer:
    def __init__(self, netG, netD_image, netD_object, netD_att, pos_weight: torch.Tensor, *, lambdas: Optional[dict] = None,
                 group=None, estimate_attributes: bool = False, reuse_generator_pass: bool = True,
                 conv_dtype: str = "f32", streams: bool = True):
        self.netG, self.netDi, self.netDo, self.netDa = netG, netD_image, netD_object, netD_att
        dev = next(netG.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("Trainer needs the networks on a HIP device (no CPU fallback)")
        self.dev = dev
        self.lam = dict(LAMBDAS, **(lambdas or {}))
        self.pos_weight = pos_weight.to(dev).float().contiguous()
        self.flat_g = FlatParams([netG])
        self.flat_d = FlatParams([netD_image, netD_object, netD_att])
        self.sync = GradSync(group)
        if self.sync.enabled:                       # identical replicas: rank 0's weights and buffers everywhere
            self.sync.broadcast_(self.flat_g.p)     # (spectral-norm u/v are randomly initialised buffers)
            self.sync.broadcast_(self.flat_d.p)
            self.flat_g.touch()
            self.flat_d.touch()
            for net in (netG, netD_image, netD_object, netD_att):
                for buf in net.buffers():
                    self.sync.broadcast_(buf)
        # True: derive attribute_est on device from the pre-step D_att logits (train64.py:156-166);
        # False (default): take the batch's attribute_est (what the parity fixtures pin).
        self.estimate_attributes = estimate_attributes
        # True: evaluate the draw-independent generator parts once per iteration (see the module docstring);
        # False: two full generator evaluations like the reference loop (A/B tests).
        self.reuse_generator_pass = reuse_generator_pass
        # arithmetic of the MFMA convolutions inside step(): "f32" exact fp32 (BASELINE config 2) or "bf16" operands with
        # fp32 accumulation (configs 3/5); passed per call through the C ABI (AGL_CONV_BF16), never a process-wide switch
        # "f32x3": fp32 tensors with fp32-accurate products on the bf16 matrix cores (three bf16 terms per operand, six
        # products, AGL_CONV_SPLIT3) in the kernels that support it, exact fp32 MFMA elsewhere
        self.conv_flags = {"f32": 0, "fp32": 0, "f32x3": L.CONV_SPLIT3, "bf16": L.CONV_BF16}[conv_dtype]
        # weight gradients on their own stream beside the input-gradient chain (agl.lib.WGRAD_STREAM); AGL_WGRAD_STREAM=0: off
        # `streams=False` (or the AGL_*_STREAMS=0 environment switches, for A/B runs): everything on the caller's stream, in
        # program order — same kernels, same arithmetic; tests/test_model_gpu.py compares the two schedules.
        if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)      # (intended: see the chains below)
        on = lambda name: streams and os.environ.get(name, "1") != "0"
        self.wgrad_stream = torch.cuda.Stream(device=dev) if on("AGL_WGRAD_STREAM") else None
        # The three discriminators are independent networks: their forward / backward chains run on three streams beside each
        # other (calls of ONE network stay in order on its stream, so its spectral-norm state advances exactly as before, and
        # its gradient slots have one writer chain).  Every convolution launch ends in a partial round of workgroups and the
        # deep layers have small grids; concurrent chains fill those holes.  AGL_D_STREAMS=0: everything on one stream.
        self.d_streams = [torch.cuda.Stream(device=dev) for _ in range(3)] if on("AGL_D_STREAMS") else None
        # ... and so do the generator's `rand` and `shift` branches behind the ConvLSTM (agl.generator.Generator.part_b)
        #     (streams 0, 1), and the reconstruction branch of the G step's evaluation (stream 2: its forward runs beside the D step,
        #     its backward — a ConvLSTM recurrence of small kernels — beside the other branches' backward)
        self.g_streams = [torch.cuda.Stream(device=dev) for _ in range(3)] if on("AGL_G_STREAMS") else None
        netG.__dict__.pop("branch_streams", None)
        netG.__dict__.pop("branch_grad_arenas", None)
        if self.g_streams is not None:
            netG.__dict__["branch_streams"] = self.g_streams
            # the branches share their layers: each accumulates its parameter gradients into a private arena (no two streams
            # read-modify-write one slot), folded into flat_g.g after the backward pass
            netG.__dict__["branch_grad_arenas"] = [(self.flat_g, t) for t in self.flat_g.branch_arenas(3)]
        self._wgrad_map = None
        if self.wgrad_stream is not None:
            self._wgrad_map = {torch.cuda.current_stream(dev).cuda_stream: self.wgrad_stream}
            for st in (self.d_streams or []) + (self.g_streams or []):
                self._wgrad_map[st.cuda_stream] = torch.cuda.Stream(device=dev)
        self._in_step = False
        # With data parallelism step() returns while the G all-reduce + Adam still run on the side stream.  Readers of
        # the weights outside step() (state_dict / checkpoint.save_model, eval or user forwards) join it first.
        for net in (netG, netD_image, netD_object, netD_att):
            net.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.finish())
            net.register_forward_pre_hook(lambda module, args: None if self._in_step else self.finish())
        self.raw = torch.zeros(len(RAW), dtype=torch.float32, device=dev)
        self._keep = {}
        self._d_ready = None
        self._g_ready = None
        self.on_d_backward = None        # optional test probes, called after backward and before Adam
        self.on_g_backward = None
        self.marks = None                # tools/phase_times.py: a list collects (label, event) pairs along the iteration

    # ------------------------------------------------------------------ helpers
    def serial(self):
        """Context manager: the iterations inside run on ONE stream in program order (as Trainer(streams=False) does) — used by
        bench.py's instrumented step, where every launch is timed alone, not beside the kernels of the other chains."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            saved = (self.wgrad_stream, self.d_streams, self.g_streams, self._wgrad_map,
                     self.netG.__dict__.pop("branch_streams", None), self.netG.__dict__.pop("branch_grad_arenas", None))
            self.finish()
            torch.cuda.synchronize()
            self.wgrad_stream = self.d_streams = self.g_streams = self._wgrad_map = None
            try:
                yield self
            finally:
                torch.cuda.synchronize()
                self.wgrad_stream, self.d_streams, self.g_streams, self._wgrad_map, bs, ba = saved
                if bs is not None:
                    self.netG.__dict__["branch_streams"], self.netG.__dict__["branch_grad_arenas"] = bs, ba
        return cm()

    def _slot(self, name):
        i = IDX[name]
        return self.raw[i:i + 1]

    def _mark(self, label, stream=None):
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self.marks.append((label, ev, time.perf_counter()))

    def _gen(self, b, eps):
        return self.netG(b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], b["z"], b["attribute"],
                         b["masks_shift"], b["boxes_shift"], b["attribute_est"], eps=eps)

    def _gen_first(self, b, eps):
        """First generator evaluation of the iteration: draw-independent parts with a graph and a BatchNorm tape,
        the reconstruction branch without a graph.  Returns (outputs, state for _gen_second)."""
        G = self.netG
        tape_a, tape_b = [], []
        try:
            F.BN_TAPE = tape_a
            sh = G.part_a(b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], b["z"], b["attribute"],
                          b["masks_shift"], b["boxes_shift"], b["attribute_est"])
            F.BN_TAPE = None
            e = G.draw_eps(sh, eps)
            bs = G.__dict__.get("branch_streams")
            if bs is not None and len(bs) >= 3 and G.batch_clstm and G.training:
                rec = G.part_rec_nograd_beside_b(sh, e[0], tape_b)      # the graph-less reconstruction branch beside part_b
            else:
                with torch.no_grad():
                    rec = G.part_rec(sh, e[0])
                F.BN_TAPE = tape_b
                G.part_b(sh)
        finally:
            F.BN_TAPE = None
        with torch.no_grad():
            out = G.outputs(sh, rec, e)
        return tuple(t.detach() for t in out), (sh, tape_a, tape_b)

    def _gen_second(self, state, eps):
        """Second evaluation: the recorded layers advance their running statistics again (same batches), the
        reconstruction branch is recomputed with the new draws, this time with a graph."""
        G = self.netG
        sh, tape_a, tape_b = state
        e = G.draw_eps(sh, eps)
        F.bn_tape_replay(tape_a)          # crop-encoder trunk on the real crops, attribute encoder
        rec = G.part_rec(sh, e[0])
        F.bn_tape_replay(tape_b)          # rand / shift branches and the two crop-encoder trunks after them
        return G.outputs(sh, rec, e)

    def _reduce_and_step(self, flat: FlatParams):
        """All-reduce the arena's gradients (side stream) and apply Adam there; returns the event the main
        stream must wait on before it reads the updated weights."""
        # (the callers have joined every chain and weight-gradient stream into the current stream: the precondition of
        #  FlatParams.adam_step and of the in-place re-pack behind it — agl.lib, pack-cache contract)
        if not self.sync.enabled:
            L.note_joined()
            flat.adam_step(LR, BETA1, BETA2, ADAM_EPS, 1.0)
            return None
        main = torch.cuda.current_stream(self.dev)
        side = self.sync.side_stream(self.dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            L.note_joined(side)
            self.sync.all_reduce_(flat.g)
            flat.adam_step(LR, BETA1, BETA2, ADAM_EPS, self.sync.grad_scale)
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    @staticmethod
    def _wait(ev):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _backward(self, heads, grads):
        """torch.autograd.backward with the weight-gradient kernels on the side stream; joined before anything reads the slots."""
        main = torch.cuda.current_stream()
        if self._wgrad_map is not None and main.cuda_stream not in self._wgrad_map:      # (a caller-chosen stream: give it a partner)
            self._wgrad_map[main.cuda_stream] = self.wgrad_stream
        L.WGRAD_STREAMS = self._wgrad_map
        try:
            torch.autograd.backward(heads, grads)
        finally:
            L.WGRAD_STREAMS = None
        for st in (self.d_streams or []) + (self.g_streams or []):      # the chains ran their backward on their own streams
            main.wait_stream(st)
        for st in (self._wgrad_map or {}).values():
            main.wait_stream(st)

    class _Chains:
        """Context of the per-network streams: `with chains.on(k): ...` runs a discriminator's calls on stream k."""

        def __init__(self, streams):
            self.streams = streams
            self.main = torch.cuda.current_stream()
            for st in streams or []:
                st.wait_stream(self.main)

        def on(self, k, *reads):
            """`reads`: tensors produced on other streams that the chain's kernels read (marked for the allocator, agl.lib.used_on)."""
            import contextlib
            if self.streams:
                L.used_on(self.streams[k], *reads)
            return torch.cuda.stream(self.streams[k]) if self.streams else contextlib.nullcontext()

        def join(self):
            for st in self.streams or []:
                self.main.wait_stream(st)

    # ------------------------------------------------------------------ the step
    def step(self, b: Dict[str, torch.Tensor], eps_d: Optional[Sequence[torch.Tensor]] = None,
             eps_g: Optional[Sequence[torch.Tensor]] = None):
        with _STEP_LOCK:
            self._in_step = True
            saved = (F.BN_TAPE, F.BN_DEFER, F.GRAD_ARENA, F.EMIT_STATS, L.WGRAD_STREAMS)
            try:
                with L.conv_flags(self.conv_flags):
                    return self._step(b, eps_d, eps_g)
            finally:
                F.BN_TAPE, F.BN_DEFER, F.GRAD_ARENA, F.EMIT_STATS, L.WGRAD_STREAMS = saved
                F._LAST_STATS = None
                self._in_step = False

    def _step(self, b, eps_d, eps_g):
        lam = self.lam
        objs = b["objs"]
        N = b["imgs"].shape[0]
        n_swap = math.floor(N / 3)                                        # train64.py:170
        o2i = b["obj_to_img"]
        if not o2i.is_cuda and o2i.numel() and (int(o2i.min()) < 0 or int(o2i.max()) >= N):
            raise IndexError(f"obj_to_img must lie in [0, {N})")
        o2i_dev = L.box_map_to_device(o2i, self.dev)
        s = self.netG.obj_size

        # ---- pre-step attribute-estimate forward (train64.py:160-161); independent of G, so the previous
        #      iteration's G all-reduce/Adam (side stream) overlaps it
        #      With the discriminator streams it runs on D_att's stream beside the generator's crop-encoder trunk; the generator
        #      waits for the estimate where it first reads it (the event travels with the tensor).
        self._mark("start")
        pre = self._Chains(self.d_streams[2:3] if self.d_streams else None)
        with torch.no_grad(), pre.on(0):
            crops_real = F.crop_boxes(b["imgs"], b["boxes"], o2i_dev, s)
            att_logits = self.netDa(crops_real)
            if self.estimate_attributes:
                est = L.attr_estimate(att_logits, b["attribute"])
                if self.d_streams:
                    est._agl_ready = torch.cuda.Event()
                    est._agl_ready.record(self.d_streams[2])
                    L.used_on(torch.cuda.current_stream(), est)      # written on D_att's stream, read by the generator on this one
                b = dict(b, attribute_est=est)
            self._mark("pre-step estimate done (its stream)")
        self._wait(self._g_ready)
        self._g_ready = None

        # ---- D step (train64.py:191-262)
        gen_state = None
        if self.reuse_generator_pass and self.netG.training:
            out, gen_state = self._gen_first(b, eps_d)
        else:
            with torch.no_grad():
                out = self._gen(b, eps_d)
        crops_input, crops_rec, crops_rand, crops_shift, img_rec, img_rand, img_shift = out[:7]
        self._mark("generator pass 1 done")
        self.flat_d.zero_grad()
        heads, grads = [], []

        def term(t, g):
            heads.append(t)
            grads.append(g)

        ch = self._Chains(self.d_streams)
        with ch.on(0, img_rec, img_rand, img_shift):
            for name, x, w in (("d_img_rec", img_rec, MIX[0]), ("d_img_rand", img_rand, MIX[1]), ("d_img_shift", img_shift, MIX[2])):
                lg = self.netDi(x)
                term(lg, LS.bce_const(lg, 0.0, lam["img_adv"] * w, self._slot(name)))
            lg = self.netDi(b["imgs"])
            term(lg, LS.bce_const(lg, 1.0, lam["img_adv"], self._slot("d_img_real")))
            self._mark("D step: D_img forward done (its stream)")
        with ch.on(1, crops_rec, crops_rand, crops_shift, crops_input):
            for name, x, w in (("d_obj_rec", crops_rec, MIX[0]), ("d_obj_rand", crops_rand, MIX[1]), ("d_obj_shift", crops_shift, MIX[2])):
                src, _ = self.netDo(x, objs)
                term(src, LS.bce_const(src, 0.0, lam["obj_adv"] * w, self._slot(name)))
            src, cls = self.netDo(crops_input, objs)
            term(src, LS.bce_const(src, 1.0, lam["obj_adv"], self._slot("d_obj_real")))
            term(cls, LS.cross_entropy(cls, objs, lam["obj_cls"], self._slot("d_obj_cls")))
            self._mark("D step: D_obj forward done (its stream)")
        with ch.on(2, crops_input):
            att = self.netDa(crops_input)
            term(att, LS.bce_posw(att, b["attribute_gt"], self.pos_weight, lam["att_cls"], self._slot("d_att")))
            self._mark("D step: D_att forward done (its stream)")
        # The G step's generator evaluation (train64.py:280) reads the generator's weights and the batch only: with the discriminator
        # chains on their own streams it is issued here, on the main stream, beside the D step's forward passes (the BatchNorm
        # updates of the first evaluation are complete; the CPU draws are consumed in the reference's order)
        out_g = None
        if gen_state is not None and self.d_streams:
            if self.g_streams and G_REC_STREAM:   # (measured slower: 460 vs 470 images/s; kept as a constant for the equivalence test)
                # (measured on one box, alternating: 460 images/s with it against 470 without — beside the D step it competes with
                #  the chains on the critical path — so the default keeps it on the main stream)
                main, g2 = torch.cuda.current_stream(), self.g_streams[2]
                g2.wait_stream(main)
                prev, F.GRAD_ARENA = F.GRAD_ARENA, self.netG.__dict__["branch_grad_arenas"][2]
                try:
                    with torch.cuda.stream(g2):
                        out_g = self._gen_second(gen_state, eps_g)
                finally:
                    F.GRAD_ARENA = prev
            else:
                out_g = self._gen_second(gen_state, eps_g)
            gen_state = None
            self._mark("generator pass 2 done (main stream)")
        ch.join()
        self._mark("D step: forward joined")
        self._backward(heads, grads)
        self._mark("D step: backward joined")
        if self.on_d_backward is not None:
            self.on_d_backward(self)
        self._d_ready = self._reduce_and_step(self.flat_d)
        self._mark("D step: Adam done")

        # ---- G step (train64.py:280-370)
        self.flat_d.set_requires_grad(False)
        try:
            # (without the discriminator streams: overlaps the D all-reduce + Adam on the side stream)
            out = out_g if out_g is not None else (self._gen_second(gen_state, eps_g) if gen_state is not None else self._gen(b, eps_g))
            gen_state = None
            if out_g is not None and self.g_streams:
                torch.cuda.current_stream().wait_stream(self.g_streams[2])      # (no-op when the stream was not used)
            (crops_input, crops_rec, crops_rand, crops_shift, img_rec, img_rand, img_shift,
             mu, logvar, z_rand_rec, z_rand_shift) = out
            self._wait(self._d_ready)
            self._d_ready = None
            self.flat_g.zero_grad()
            heads, grads = [], []
            keep = self._keep.get(N)                                       # (constant per batch size: built once, two launches less per step)
            if keep is None:
                keep = torch.ones(N, dtype=torch.float32, device=self.dev)
                keep[:n_swap] = 0                                          # :284
                self._keep[N] = keep
            term(img_rec, LS.l1_rows(img_rec, b["imgs"], keep, lam["img_rec"], float(N - n_swap), self._slot("g_img_rec")))
            term(z_rand_rec, LS.l1_rows(z_rand_rec, b["z"], None, lam["z_rec"] * 0.5, 1.0, self._slot("g_z_rand")))
            term(z_rand_shift, LS.l1_rows(z_rand_shift, b["z"], None, lam["z_rec"] * 0.5, 1.0, self._slot("g_z_shift")))
            dmu, dlv = LS.kl_sum(mu, logvar, lam["kl"], self._slot("g_kl"))
            term(mu, dmu)
            term(logvar, dlv)
            ch = self._Chains(self.d_streams)
            with ch.on(0, img_rec, img_rand, img_shift):
                for tag, x, w in (("rec", img_rec, MIX[0]), ("rand", img_rand, MIX[1]), ("shift", img_shift, MIX[2])):
                    lg = self.netDi(x)
                    term(lg, LS.bce_const(lg, 1.0, lam["img_adv"] * w, self._slot("g_img_adv_" + tag)))
                self._mark("G step: D_img forward done (its stream)")
            with ch.on(1, crops_rec, crops_rand, crops_shift):
                for tag, x, w in (("rec", crops_rec, MIX[0]), ("rand", crops_rand, MIX[1]), ("shift", crops_shift, MIX[2])):
                    src, cls = self.netDo(x, objs)
                    term(src, LS.bce_const(src, 1.0, lam["obj_adv"] * w, self._slot("g_obj_adv_" + tag)))
                    term(cls, LS.cross_entropy(cls, objs, lam["obj_cls"] * w, self._slot("g_obj_cls_" + tag)))
                self._mark("G step: D_obj forward done (its stream)")
            with ch.on(2, crops_rec, crops_rand, crops_shift):
                for tag, x, w in (("rec", crops_rec, MIX[0]), ("rand", crops_rand, MIX[1]), ("shift", crops_shift, MIX[2])):
                    att = self.netDa(x)
                    term(att, LS.bce_posw(att, b["attribute"], self.pos_weight, lam["att_cls"] * w, self._slot("g_att_" + tag)))
                self._mark("G step: D_att forward done (its stream)")
            ch.join()
            self._mark("G step: forward joined")
            self._backward(heads, grads)
            self._mark("G step: backward joined")
            self.flat_g.fold_branch_arenas()
            if self.on_g_backward is not None:
                self.on_g_backward(self)
        finally:
            self.flat_d.set_requires_grad(True)
        self._g_ready = self._reduce_and_step(self.flat_g)
        self._mark("G step: Adam done")
        self.last_outputs = out
        return self.raw

    def finish(self):
        """Join the side stream (call before reading weights / at the end of a timed region)."""
        self._wait(self._g_ready)
        self._wait(self._d_ready)
        self._g_ready = self._d_ready = None

    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        """Adam moments and step counts of both arenas (what torch.optim.Adam.state_dict() would carry)."""
        self.finish()
        return {"g_m": self.flat_g.m.clone(), "g_v": self.flat_g.v.clone(), "g_step": self.flat_g.step_count,
                "d_m": self.flat_d.m.clone(), "d_v": self.flat_d.v.clone(), "d_step": self.flat_d.step_count}

    def load_optimizer_state(self, st):
        self.finish()
        for flat, k in ((self.flat_g, "g"), (self.flat_d, "d")):
            if st[k + "_m"].numel() != flat.n:
                raise ValueError(f"optimizer state size {st[k + '_m'].numel()} != arena size {flat.n}")
            flat.m.copy_(st[k + "_m"])
            flat.v.copy_(st[k + "_v"])
            flat.step_count = int(st[k + "_step"])

    def loss_dict(self) -> Dict[str, float]:
        """The 15 scalars the reference logs (train64.py:266-272, :372-379); one device->host copy."""
        r = {k: float(v) for k, v in zip(RAW, self.raw.detach().cpu().tolist())}
        lam = self.lam
        mix = lambda p: MIX[0] * r[p + "_rec"] + MIX[1] * r[p + "_rand"] + MIX[2] * r[p + "_shift"]
        d_img_fake, d_obj_fake = mix("d_img"), mix("d_obj")
        g_img_adv, g_obj_adv, g_obj_cls, g_att = mix("g_img_adv"), mix("g_obj_adv"), mix("g_obj_cls"), mix("g_att")
        g_z = 0.5 * r["g_z_rand"] + 0.5 * r["g_z_shift"]
        d_loss = (lam["img_adv"] * (d_img_fake + r["d_img_real"]) + lam["obj_adv"] * (d_obj_fake + r["d_obj_real"])
                  + lam["obj_cls"] * r["d_obj_cls"] + lam["att_cls"] * r["d_att"])
        g_loss = (lam["img_rec"] * r["g_img_rec"] + lam["z_rec"] * g_z + lam["img_adv"] * g_img_adv + lam["obj_adv"] * g_obj_adv
                  + lam["obj_cls"] * g_obj_cls + lam["att_cls"] * g_att + lam["kl"] * r["g_kl"])
        return {"D/loss": d_loss, "D/image_adv_loss_real": r["d_img_real"], "D/image_adv_loss_fake": d_img_fake,
                "D/object_adv_loss_real": r["d_obj_real"], "D/object_adv_loss_fake": d_obj_fake,
                "D/object_cls_loss_real": r["d_obj_cls"], "D/object_att_cls_loss": r["d_att"],
                "G/loss": g_loss, "G/image_adv_loss": g_img_adv, "G/object_adv_loss": g_obj_adv,
                "G/object_cls_loss": g_obj_cls, "G/rec_img": r["g_img_rec"], "G/rec_z": g_z, "G/kl": r["g_kl"],
                "G/object_att_cls_loss": g_att}
