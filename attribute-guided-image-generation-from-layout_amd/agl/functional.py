"""Differentiable ops of the G+D path: torch.autograd.Function shells around the HIP kernels.

autograd is used as plumbing only (graph bookkeeping); every forward and backward body is a
sequence of libagl.so launches (agl.lib).  Reference call sites are cited per op.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import lib as L

BN_EPS, BN_MOMENTUM, SN_EPS = 1e-5, 0.1, 1e-12


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# Tape of BatchNorm statistic updates (agl.generator's two-pass schedule): while BN_TAPE is a list, every training-mode
# statistics call appends a closure that re-applies exactly the same running-statistics update (it re-runs the cheap
# statistics kernel on the saved input), so a later pass can reuse the recorded activations and still advance the
# running statistics as if the layers had been evaluated again.
BN_TAPE = None


def bn_tape_replay(tape):
    """Entries: ("bn", moments, running_mean, running_var, nbt) — a running-statistics update, replayed from its saved moments;
    runs of them go out as one launch — or a callable."""
    run = []
    cur = torch.cuda.current_stream()
    for e in tape:
        if isinstance(e, tuple):
            L.used_on(cur, e[1])      # (the moments may have been written on a branch stream)
            run.append(e[1:])
            continue
        if run:
            L.bn_running_update_many(run, BN_MOMENTUM)
            run = []
        e()
    if run:
        L.bn_running_update_many(run, BN_MOMENTUM)


# Deferred running-statistics updates (agl.generator's concurrent branches): while BN_DEFER is a list, a training-mode statistics
# call leaves its moments and does NOT touch the running statistics; bn_apply_deferred performs the updates later, in list order.
# Two branches that share BatchNorm layers can then be evaluated side by side on two streams and still update every layer's
# running statistics in the reference's order (first branch, then second).
BN_DEFER = None


def bn_apply_deferred(entries):
    if entries:
        cur = torch.cuda.current_stream()
        for e in entries:
            L.used_on(cur, e[0])      # moments written on a branch stream, read here
        L.bn_running_update_many(list(entries), BN_MOMENTUM)


def _slot(p):
    """Gradient slot of a parameter owned by a flat arena (agl.flat.FlatParams marks them): backward kernels then
    accumulate straight into the arena and return None, instead of handing autograd a temporary that it adds to
    .grad with one extra elementwise launch per parameter per use (~700 launches per training iteration)."""
    if p is None or not getattr(p, "_agl_slot", False) or p.grad is None:
        return None
    a = GRAD_ARENA
    if a is not None and getattr(p, "_agl_flat", None) is a[0]:
        off = p._agl_off
        return a[1][off:off + p.numel()].view(p.shape)
    return p.grad


# Private gradient arena of a concurrent branch: (FlatParams, tensor laid out like its .g) or None.  Two generator branches that run
# on two streams share their layers; their backward kernels would read-modify-write the same gradient slots at the same time.  While
# GRAD_ARENA is set (during a branch's FORWARD: the slots are chosen there), the slots of that arena's parameters point into the
# branch's own buffer instead; agl.trainer adds the buffers to the arena's gradient after the backward pass.
GRAD_ARENA = None



# BatchNorm statistics produced by the convolution that feeds the norm (agl_conv2d_fwd_stats): while EMIT_STATS is set (the
# generator's forward: every conv there is followed by a batch-statistics norm), a plain convolution also leaves per-channel
# partial sums of its output, and the _NormAct that consumes exactly that tensor next finalises them instead of re-reading
# the activation.  One entry at most, dropped by whatever convolution or norm runs next.
EMIT_STATS = False
_LAST_STATS = None      # (y, y._version at emission, partials, rows): the entry holds y itself, so its storage cannot be recycled
                        # for another tensor while the entry is alive, and an in-place write to y invalidates it


def _take_stats(x):
    global _LAST_STATS
    e, _LAST_STATS = _LAST_STATS, None
    if e is not None and e[0] is x and e[1] == x._version:
        return e[2], e[3]
    return None, 0


# --------------------------------------------------------------------------- convolution
def _conv_param_grads(slots, want_w, want_b, g, x, w, stride, pad, up, in_relu):
    """Weight and bias gradient of a convolution (shared by _Conv2d and _ConvPoolFork): into the arena slots of the parameters
    when they have them (returns None for those), on the chain's weight-gradient stream."""
    wslot, bslot = slots
    ks = w.shape[2]
    dw = db = None
    if want_w:
        # the bias gradient rides on the weight-gradient kernel (it stages dy anyway) when both go the same way
        # (both into their arena slots, or both into fresh tensors)
        if wslot is not None:
            fuse_b = want_b and bslot is not None
            L.on_wgrad_stream(lambda: L.conv2d_bwd_weight(g, x, ks, stride, pad, up, in_relu, out=wslot, accumulate=True,
                                                          dbias=bslot if fuse_b else None), g, x)
        else:      # (dw to a fresh tensor — e.g. a spectrally normalised weight; the bias may still have its arena slot)
            fuse_b = want_b
            if fuse_b and bslot is None:
                db = torch.empty(w.shape[0], dtype=torch.float32, device=g.device)
            dw = L.conv2d_bwd_weight(g, x, ks, stride, pad, up, in_relu, dbias=(bslot if bslot is not None else db) if fuse_b else None,
                                     dbias_accumulate=bslot is not None)
        want_b = want_b and not fuse_b
    if want_b:
        if bslot is not None:
            L.channel_sum(g, out=bslot, accumulate=True)
        else:
            db = L.channel_sum(g)
    return dw, db


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, addend, stride, pad, up, in_relu, relu, relu_grad_by_consumer, x_relu):
        global _LAST_STATS
        _LAST_STATS = None
        ctx.slots = (_slot(w), _slot(bias))
        ctx.wsrc = wsrc = getattr(w, "_agl_wsrc", None)        # packed-weight cache handle (agl.lib.WeightSrc), if the weight has one
        x, w = _c(x), _c(w)
        if addend is not None:
            assert not relu, "addend with fused ReLU is not supported"
            y = L.conv2d_fwd(x, w, bias, stride, pad, up, in_relu, False, out=addend, accumulate=True, wsrc=wsrc)
            ctx.mark_dirty(addend)
        elif EMIT_STATS and not relu and (L.CONV_FLAGS & (L.CONV_BF16 | L.CONV_SPLIT3)):
            y, part, rows = L.conv2d_fwd_stats(x, w, bias, stride, pad, up, in_relu, wsrc=wsrc)
            if part is not None:
                _LAST_STATS = (y, y._version, part, rows)
        else:
            y = L.conv2d_fwd(x, w, bias, stride, pad, up, in_relu, relu, wsrc=wsrc)
        ctx.cfg = (stride, pad, up, in_relu, relu, bias is not None, addend is not None, relu_grad_by_consumer, x_relu)
        ctx.save_for_backward(x, w, y if (relu and not relu_grad_by_consumer) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, pad, up, in_relu, relu, has_bias, has_add, by_consumer, x_relu = ctx.cfg
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        # relu_grad_by_consumer: the (single) consumer of y masks its input gradient with y > 0 (its x_relu flag), so
        # dy already is the gradient of the pre-activation and the separate masking pass is skipped
        g = L.relu_bwd(dy, y) if (relu and not by_consumer) else dy
        ks = w.shape[2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if up:
                H, W = x.shape[2] << up, x.shape[3] << up
                dxu = L.conv2d_bwd_data(g, w, (H, W), stride, pad, wsrc=ctx.wsrc)
                dx = L.upsample_bwd(dxu, up)
                assert not in_relu
            else:
                dx = L.conv2d_bwd_data(g, w, (x.shape[2], x.shape[3]), stride, pad, pos_mask=x if (in_relu or x_relu) else None,
                                       wsrc=ctx.wsrc)
        dw, db = _conv_param_grads(ctx.slots, ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2], g, x, w, stride, pad, up, in_relu)
        dadd = dy if (has_add and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dadd, None, None, None, None, None, None, None


def conv2d(x, w, bias=None, stride=1, padding=0, up=0, in_relu=False, relu=False, addend=None, relu_grad_by_consumer=False,
           x_relu=False):
    """nn.Conv2d forward (+ optional fused input ReLU / nearest up-sampling of x, output ReLU, and
    accumulation into `addend`, which is consumed in place).

    A producer/consumer pair can drop the separate ReLU-backward pass: the producer passes relu=True,
    relu_grad_by_consumer=True and its output must feed exactly ONE convolution, called with x_relu=True, which masks
    the gradient it returns by x > 0 in its input-gradient epilogue."""
    assert not (x_relu and up), "x_relu with folded up-sampling is not supported"
    return _Conv2d.apply(x, w, bias, addend, stride, padding, up, in_relu, relu, relu_grad_by_consumer, x_relu)


class _ConvPoolFork(torch.autograd.Function):
    """The two consumers of a down-sampling residual block's input (reference discriminator.py:46-60, 84-99) as ONE graph node:
    y = conv(x) (3x3, stride 1, optional fused input / output ReLU) and s = avg_pool2d(x, 2) (of relu(x) with in_relu: the block's
    in-place ReLU aliases the shortcut's input).  Same kernels as conv2d + avg_pool2 forward; backward: the pooled gradient is
    written to dx and the convolution's input-gradient epilogue accumulates onto it — autograd's separate add of two x-sized
    tensors (3 passes over them) disappears."""

    @staticmethod
    def forward(ctx, x, w, bias, pad, in_relu, relu, relu_grad_by_consumer):
        global _LAST_STATS
        _LAST_STATS = None
        ctx.slots = (_slot(w), _slot(bias))
        ctx.wsrc = wsrc = getattr(w, "_agl_wsrc", None)
        x, w = _c(x), _c(w)
        y = L.conv2d_fwd(x, w, bias, 1, pad, 0, in_relu, relu, wsrc=wsrc)
        s = L.avgpool2_fwd(x, in_relu)
        ctx.cfg = (pad, in_relu, relu, bias is not None, relu_grad_by_consumer)
        ctx.save_for_backward(x, w, y if (relu and not relu_grad_by_consumer) else None)
        return y, s

    @staticmethod
    def backward(ctx, dy, ds):
        pad, in_relu, relu, has_bias, by_consumer = ctx.cfg
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        g = L.relu_bwd(dy, y) if (relu and not by_consumer) else dy
        dx = None
        if ctx.needs_input_grad[0]:
            dx = L.avgpool2_bwd(_c(ds), x if in_relu else tuple(x.shape), in_relu)
            L.conv2d_bwd_data(g, w, (x.shape[2], x.shape[3]), 1, pad, pos_mask=x if in_relu else None, out=dx, accumulate=True,
                              wsrc=ctx.wsrc)
        dw, db = _conv_param_grads(ctx.slots, ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2], g, x, w, 1, pad, 0, in_relu)
        return dx, dw, db, None, None, None, None


def conv2d_and_avg_pool2(x, w, bias=None, padding=1, in_relu=False, relu=False, relu_grad_by_consumer=False):
    """(conv2d(x, w, bias, stride 1, padding), avg_pool2(x)) — see _ConvPoolFork."""
    return _ConvPoolFork.apply(x, w, bias, padding, in_relu, relu, relu_grad_by_consumer)


class _ConvReluConv(torch.autograd.Function):
    """y = conv3x3(relu(conv_k(x, w1, b1)), w2, b2), both stride 1 / "same" (the residual branch of a discriminator block without
    down-sampling, reference discriminator.py:36-40) as ONE graph node, so that the intermediate h = relu(conv_k(x)) never is an
    autograd edge: in bf16 arithmetic it is then stored as bf16 (an edge of that dtype would have its gradient cast to bf16 by
    autograd).  h is read by the second convolution only — forward, the x operand of its weight gradient, the ReLU mask of its input
    gradient — which rounds it to bf16 when staging it anyway: identical results at half the traffic on that tensor.
    The same launches as two conv2d nodes with relu_grad_by_consumer / x_relu otherwise."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, pad1, h_bf16):
        global _LAST_STATS
        _LAST_STATS = None
        ctx.slots = (_slot(w1), _slot(b1), _slot(w2), _slot(b2))
        ctx.wsrcs = (getattr(w1, "_agl_wsrc", None), getattr(w2, "_agl_wsrc", None))
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        h = L.conv2d_fwd(x, w1, b1, 1, pad1, 0, False, True, wsrc=ctx.wsrcs[0], out_bf16=h_bf16)
        y = L.conv2d_fwd(h, w2, b2, 1, 1, wsrc=ctx.wsrcs[1])
        ctx.cfg = (pad1, b1 is not None, b2 is not None)
        ctx.save_for_backward(x, w1, w2, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        pad1, has_b1, has_b2 = ctx.cfg
        x, w1, w2, h = ctx.saved_tensors
        dy = _c(dy)
        need = ctx.needs_input_grad
        dx = None
        dh = None
        if need[0] or need[1] or (has_b1 and need[2]):
            dh = L.conv2d_bwd_data(dy, w2, (h.shape[2], h.shape[3]), 1, 1, pos_mask=h, wsrc=ctx.wsrcs[1])
        dw2, db2 = _conv_param_grads(ctx.slots[2:], need[3], has_b2 and need[4], dy, h, w2, 1, 1, 0, False)
        dw1 = db1 = None
        if dh is not None:
            if need[0]:
                dx = L.conv2d_bwd_data(dh, w1, (x.shape[2], x.shape[3]), 1, pad1, wsrc=ctx.wsrcs[0])
            dw1, db1 = _conv_param_grads(ctx.slots[:2], need[1], has_b1 and need[2], dh, x, w1, 1, pad1, 0, False)
        return dx, dw1, db1, dw2, db2, None, None


def conv_relu_conv3x3(x, w1, b1, w2, b2, pad1, h_bf16=False):
    return _ConvReluConv.apply(x, w1, b1, w2, b2, pad1, h_bf16)


def linear(x, w, bias=None, relu=False):
    """nn.Linear as a 1x1 convolution over (rows, features, 1, 1)."""
    y = conv2d(x.reshape(x.shape[0], x.shape[1], 1, 1), w.reshape(w.shape[0], w.shape[1], 1, 1), bias, relu=relu)
    return y.reshape(y.shape[0], y.shape[1])


class _ConvT4s2(torch.autograd.Function):
    """nn.ConvTranspose2d(k=4, s=2, p=1, bias=False) (generator_obj_att.py:532-540): the forward is the
    input-gradient pass of the matching stride-2 convolution, run one 2x2-tap phase per output parity."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.slot = _slot(w)
        ctx.wsrc = getattr(w, "_agl_wsrc", None)
        x, w = _c(x), _c(w)
        ctx.save_for_backward(x, w)
        return L.conv2d_bwd_data(x, w, (2 * x.shape[2], 2 * x.shape[3]), 2, 1, wsrc=ctx.wsrc)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = L.conv2d_fwd(dy, w, None, 2, 1, wsrc=ctx.wsrc) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if ctx.slot is not None:
                L.on_wgrad_stream(lambda: L.conv2d_bwd_weight(x, dy, 4, 2, 1, out=ctx.slot, accumulate=True), x, dy)
            else:
                dw = L.conv2d_bwd_weight(x, dy, 4, 2, 1)
        return dx, dw


def conv_transpose2d_k4s2p1(x, w):
    return _ConvT4s2.apply(x, w)


# --------------------------------------------------------------------------- normalisation
def _batch_statistics(x, rmean, rvar, nbt, training):
    """(mean, rstd) of a BatchNorm over x: batch statistics (+ the running update, taped / deferred as the schedule asks) in
    training mode — from the partial rows the producing convolution left, when it did — else the running statistics."""
    part, rows = _take_stats(x)
    if not training:
        return L.bn_stats_eval(rmean, rvar, BN_EPS)
    # on a tape the call also leaves its (mean, unbiased variance) in double: the replay re-applies the running update
    # from them — the same arithmetic on the same numbers, without a second read of the activation
    defer = BN_DEFER is not None and rmean is not None
    taped = BN_TAPE is not None and rmean is not None
    mom = torch.empty(2 * x.shape[1], dtype=torch.float64, device=x.device) if (taped or defer) else None
    upd = (None, None, None) if defer else (rmean, rvar, nbt)      # deferred: the kernel leaves the running statistics alone
    if part is not None:
        mean, rstd = L.bn_stats_from_partials(part, rows, x.shape[1], x.numel() // x.shape[1], BN_EPS, BN_MOMENTUM, *upd, moments=mom)
    else:
        mean, rstd = L.bn_stats(x, BN_EPS, BN_MOMENTUM, *upd, moments=mom)
    if defer:
        BN_DEFER.append((mom, rmean, rvar, nbt))
    if taped:
        BN_TAPE.append(("bn", mom, rmean, rvar, nbt))
    return mean, rstd


class _NormConv(torch.autograd.Function):
    """conv2d(relu?(norm(x)), w, bias) with the normalise-modulate folded into the convolution's input staging (BASELINE north_star;
    SURVEY a2): norm = BatchNorm2d without / with affine parameters (mode 0 / 1) or ConditionalBatchNorm2d (mode 2,
    generator_obj_att.py:31-44).  The normalised tensor is never stored: the forward is statistics (from the producer's partial rows
    where it left them) -> table kernel -> agl_conv2d_fwd_fold; the backward is the convolution's input gradient -> agl_norm_bwd_fold
    (ReLU mask recomputed from x) and agl_conv2d_bwd_weight_fold (same transform on the raw x).  NORM_FOLD / conv_fold_ok say when."""

    @staticmethod
    def forward(ctx, x, p0, p1, labels, rmean, rvar, nbt, mode, relu, training, w, bias, stride, pad):
        global _LAST_STATS
        ctx.pslots = (_slot(p0), _slot(p1))
        ctx.wslots = (_slot(w), _slot(bias))
        ctx.wsrc = wsrc = getattr(w, "_agl_wsrc", None)
        x, w = _c(x), _c(w)
        mean, rstd = _batch_statistics(x, rmean, rvar, nbt, training)      # (consumes / clears _LAST_STATS)
        p0c = _c(p0) if p0 is not None else None
        fold = L.norm_fold_table(mean, rstd, mode, p0c, p1, labels, x.shape[0])
        y, part, rows = L.conv2d_fwd_fold(x, fold, w, bias, stride, pad, relu, wsrc=wsrc, want_stats=EMIT_STATS)
        _LAST_STATS = (y, y._version, part, rows) if part is not None else None
        ctx.cfg = (mode, relu, training, stride, pad, bias is not None)
        ctx.fold = fold
        ctx.save_for_backward(x, mean, rstd, p0c, p1, labels, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        mode, relu, training, stride, pad, has_bias = ctx.cfg
        x, mean, rstd, p0, p1, labels, w = ctx.saved_tensors
        fold = ctx.fold
        dy = _c(dy)
        need = ctx.needs_input_grad
        ks = w.shape[2]
        dx = dp0 = dp1 = dw = db = None
        # weight (and bias) gradient: the same transform on the raw x while it is staged
        wslot, bslot = ctx.wslots
        want_b = has_bias and need[11]
        if need[10]:
            if wslot is not None:
                fuse_b = want_b and bslot is not None
                L.on_wgrad_stream(lambda: L.conv2d_bwd_weight_fold(dy, x, fold, ks, stride, pad, relu, out=wslot, accumulate=True,
                                                                   dbias=bslot if fuse_b else None), dy, x, fold.scale, fold.shift, mean)
            else:
                fuse_b = want_b
                if fuse_b and bslot is None:
                    db = torch.empty(w.shape[0], dtype=torch.float32, device=dy.device)
                dw = L.conv2d_bwd_weight_fold(dy, x, fold, ks, stride, pad, relu, dbias=(bslot if bslot is not None else db) if fuse_b else None,
                                              dbias_accumulate=bslot is not None)
            want_b = want_b and not fuse_b
        if want_b:
            if bslot is not None:
                L.channel_sum(dy, out=bslot, accumulate=True)
            else:
                db = L.channel_sum(dy)
        if need[0] or need[1] or need[2]:
            g = L.conv2d_bwd_data(dy, w, (x.shape[2], x.shape[3]), stride, pad, wsrc=ctx.wsrc)      # w.r.t. the (never stored) activation
            s0, s1 = ctx.pslots
            in_slots = False
            if mode == 1:
                if s0 is not None and s1 is not None and need[1] and need[2]:
                    dp0, dp1, in_slots = s0, s1, True
                else:
                    dp0, dp1 = torch.empty_like(p0), torch.empty_like(p1)
            elif mode == 2:
                if s0 is not None and need[1]:
                    dp0, in_slots = s0, True
                else:
                    dp0 = torch.zeros_like(p0)
            dx = L.norm_bwd_fold(g, x, mean, rstd, fold, mode, p0, p1, labels, relu, training, dp0, dp1, param_accumulate=in_slots)
            if in_slots:
                dp0 = dp1 = None
        return dx, dp0, dp1, None, None, None, None, None, None, None, dw, db, None, None


NORM_FOLD = True      # False (tests flip it in process) = normalise-modulate as its own pass in front of the convolution


def norm_conv2d(x, norm, labels, conv, relu=True, training=True):
    """conv(relu?(norm(x))) for norm = agl.nn.BatchNorm2d (affine or not) or a ConditionalBatchNorm2d-like module (.bn, .embed) and
    conv = agl.nn.Conv2d: folded into one convolution where the matrix-core kernels take the shape, else the two passes."""
    bn = getattr(norm, "bn", norm)
    if hasattr(norm, "embed"):
        mode, p0, p1 = 2, norm.embed.weight, None
    elif bn.affine:
        mode, p0, p1 = 1, bn.weight, bn.bias
    else:
        mode, p0, p1 = 0, None, None
    N, Cin, H, W = x.shape
    ks, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    if NORM_FOLD and L.conv_fold_ok(N, Cin, H, W, conv.out_channels, ks, stride, pad, need_bww=conv.weight.requires_grad):
        return _NormConv.apply(x, p0, p1, labels if mode == 2 else None, bn.running_mean, bn.running_var, bn.num_batches_tracked, mode, relu,
                               training, conv.weight, conv.bias, stride, pad)
    h = _NormAct.apply(x, p0, p1, None, labels if mode == 2 else None, bn.running_mean, bn.running_var, bn.num_batches_tracked, mode, relu, training)
    return conv(h)


def _spade_backward(g, x, y, mean, rstd, gb, relu, training, gather):
    """Backward of the SPADE modulation (mode 3): (dx, d gamma|beta on gb's grid)."""
    fused_reduce = gather is not None and x.shape[-1] in (64, 128) and x.shape[-2] == x.shape[-1]
    dgb = (torch.empty_like(gb) if fused_reduce else
           torch.empty((x.shape[0], 2 * x.shape[1]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device))
    dx = L.norm_bwd(g, x, y, mean, rstd, 3, gb, None, None, relu, training, dgb, None,
                    gb_map=gather[0] if gather is not None else None, gb_lo=gather[1] if fused_reduce else None)
    if gather is not None and not fused_reduce:
        lo, h = gather[1], gb.shape[-1]
        dsrc = torch.empty_like(gb)
        L.call("agl_grid_gather_bwd", L.ptr(dgb), L.ptr(lo, torch.int32), L.ptr(lo, torch.int32), L.ptr(dsrc), dgb.shape[0] * dgb.shape[1],
               h, h, dgb.shape[-1], dgb.shape[-1], L.stream())
        dgb = dsrc
    return dx, dgb


class _SpadeThenConv(torch.autograd.Function):
    """SPADE's normalise-modulate(+ReLU) (normalization.py:97,106) and the ONE layer that reads its output — a convolution
    (generator_obj_att128.py:588-597: spade_4 -> c6) or a ConvTranspose2d(4, 2, 1) (generator_obj_att.py:546-572: spade_k -> dc_{k+1}) —
    as one graph node, so that the modulated tensor never is an autograd edge: in bf16 arithmetic it is stored as bf16 (an edge of that
    dtype would have its gradient cast by autograd).  Its readers — the consumer's forward, the consumer's weight gradient, the ReLU
    mask of the modulation's backward — round it to bf16 when they stage it anyway: identical results at half the bytes on the largest
    tensors of the decoder.  Same launches as spade_modulate + conv2d / conv_transpose2d_k4s2p1 otherwise."""

    @staticmethod
    def forward(ctx, x, gb, rmean, rvar, nbt, relu, training, gather, w, bias, kind, stride, pad):
        global _LAST_STATS
        ctx.wslots = (_slot(w), _slot(bias))
        ctx.wsrc = wsrc = getattr(w, "_agl_wsrc", None)
        x, gb, w = _c(x), _c(gb), _c(w)
        mean, rstd = _batch_statistics(x, rmean, rvar, nbt, training)
        y16 = L.norm_apply_fwd(x, mean, rstd, 3, gb, None, None, None, relu, gb_map=gather[0] if gather is not None else None, out_bf16=True)
        _LAST_STATS = None
        if kind == "convT":
            out = L.conv2d_bwd_data(y16, w, (2 * x.shape[2], 2 * x.shape[3]), 2, 1, wsrc=wsrc)
        elif EMIT_STATS:
            out, part, rows = L.conv2d_fwd_stats(y16, w, bias, stride, pad, wsrc=wsrc)
            if part is not None:
                _LAST_STATS = (out, out._version, part, rows)
        else:
            out = L.conv2d_fwd(y16, w, bias, stride, pad, wsrc=wsrc)
        ctx.cfg = (relu, training, gather, kind, stride, pad, bias is not None)
        ctx.save_for_backward(x, y16, mean, rstd, gb, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        relu, training, gather, kind, stride, pad, has_bias = ctx.cfg
        x, y16, mean, rstd, gb, w = ctx.saved_tensors
        dout = _c(dout)
        need = ctx.needs_input_grad
        dx = dgb = dw = db = None
        g = None
        if need[0] or need[1]:
            if kind == "convT":
                g = L.conv2d_fwd(dout, w, None, 2, 1, wsrc=ctx.wsrc)
            else:
                g = L.conv2d_bwd_data(dout, w, (x.shape[2], x.shape[3]), stride, pad, wsrc=ctx.wsrc)
        if kind == "convT":
            if need[8]:
                wslot = ctx.wslots[0]
                if wslot is not None:
                    L.on_wgrad_stream(lambda: L.conv2d_bwd_weight(y16, dout, 4, 2, 1, out=wslot, accumulate=True), y16, dout)
                else:
                    dw = L.conv2d_bwd_weight(y16, dout, 4, 2, 1)
        else:
            dw, db = _conv_param_grads(ctx.wslots, need[8], has_bias and need[9], dout, y16, w, stride, pad, 0, False)
        if g is not None:
            dx, dgb = _spade_backward(g, x, y16, mean, rstd, gb, relu, training, gather)
        return dx, dgb, None, None, None, None, None, None, dw, db, None, None, None


class _SpadeFoldConv(torch.autograd.Function):
    """conv(relu?(SPADE(x))) with SPADE's normalise-modulate(+ReLU) applied by the convolution's own staging pass (include/agl.h
    agl_conv2d_fwd_spade; BASELINE north_star "SPADE normalization fused with the following conv"): the modulated tensor of
    normalization.py:97,106 is never written — the forward, the weight gradient and the backward's ReLU mask each evaluate
    csrc/spade.h's expression on the raw x.  Taken where gamma|beta live on a class grid much smaller than the map (the 128 px
    decoder's spade_4 -> c6, spade_5 -> c7, generator_obj_att128.py:588-597) in bf16 arithmetic; bit-identical to _SpadeThenConv."""

    @staticmethod
    def forward(ctx, x, gb, rmean, rvar, nbt, relu, training, gather, w, bias, stride, pad):
        global _LAST_STATS
        ctx.wslots = (_slot(w), _slot(bias))
        ctx.wsrc = wsrc = getattr(w, "_agl_wsrc", None)
        x, gb, w = _c(x), _c(gb), _c(w)
        mean, rstd = _batch_statistics(x, rmean, rvar, nbt, training)
        sp = L.SpadeFold(mean, rstd, gb, gather[0])
        _LAST_STATS = None
        out, part, rows = L.conv2d_fwd_spade(x, sp, w, bias, stride, pad, in_relu=relu, wsrc=wsrc, want_stats=EMIT_STATS)
        if part is not None:
            _LAST_STATS = (out, out._version, part, rows)
        ctx.cfg = (relu, training, gather, stride, pad, bias is not None)
        ctx.save_for_backward(x, mean, rstd, gb, w, sp.cells)
        return out

    @staticmethod
    def backward(ctx, dout):
        relu, training, gather, stride, pad, has_bias = ctx.cfg
        x, mean, rstd, gb, w, cells = ctx.saved_tensors
        dout = _c(dout)
        need = ctx.needs_input_grad
        dx = dgb = dw = db = None
        sp = L.SpadeFold.__new__(L.SpadeFold)
        sp.mean, sp.rstd, sp.cells, sp.map, sp.G = mean, rstd, cells, gather[0], gb.shape[-1]
        ks = w.shape[2]
        g = L.conv2d_bwd_data(dout, w, (x.shape[2], x.shape[3]), stride, pad, wsrc=ctx.wsrc) if (need[0] or need[1]) else None
        wslot, bslot = ctx.wslots
        want_b = has_bias and need[9]
        if need[8]:
            if wslot is not None:
                fuse_b = want_b and bslot is not None
                L.on_wgrad_stream(lambda: L.conv2d_bwd_weight_spade(dout, x, sp, ks, stride, pad, relu, out=wslot, accumulate=True,
                                                                    dbias=bslot if fuse_b else None), dout, x)
            else:
                fuse_b = want_b
                if fuse_b and bslot is None:
                    db = torch.empty(w.shape[0], dtype=torch.float32, device=dout.device)
                dw = L.conv2d_bwd_weight_spade(dout, x, sp, ks, stride, pad, relu, dbias=(bslot if bslot is not None else db) if fuse_b else None,
                                               dbias_accumulate=bslot is not None)
            want_b = want_b and not fuse_b
        if want_b:
            if bslot is not None:
                L.channel_sum(dout, out=bslot, accumulate=True)
            else:
                db = L.channel_sum(dout)
        if g is not None:
            dgb = torch.empty_like(gb)
            dx = L.norm_bwd_spade(g, x, mean, rstd, gb, relu, training, dgb, gb_map=gather[0], gb_lo=gather[1])
        return dx, dgb, None, None, None, None, None, None, dw, db, None, None


# Off by default: measured SLOWER at config 3 (profiles/r05_ab_spade_fold.txt: +1.8 ms of 124.6 ms serial kernel time, 288-290 against
# 291-296 images/s) — the apply pass it removes costs 0.74 ms, while the consumers' staging passes, which are bound by load issue, pay
# 8 four-byte + 4 sixteen-byte loads per 8-channel item in place of 8 two-byte loads, un-prefetched in the two weight-gradient kernels.
SPADE_FOLD = False     # True: SPADE's modulate + ReLU applied by the staging pass of the convolution that reads it (_SpadeFoldConv)
SPADE_Y16 = True      # False (tests flip it in process) keeps the modulated tensors in fp32 in bf16 arithmetic


def spade_modulate_then(x, gb, rmean, rvar, nbt, relu, training, gather, consumer):
    """spade_modulate(...) followed by `consumer` (agl.nn.Conv2d or agl.nn.ConvTranspose2d), with the modulated tensor stored as
    bf16 between them where bf16 arithmetic and the kernels allow (_SpadeThenConv); else the two graph nodes."""
    g = None
    if gather is not None:
        m, lo, src = _grid_map(gather[0], gather[1], gather[2], x.device)
        assert gb.shape[2] == gb.shape[3] == src and x.shape[2] == x.shape[3] == m.numel(), (gb.shape, x.shape, gather)
        g = (m, lo)
    N, Cc, H, W = x.shape
    is_t = isinstance(consumer, torch.nn.ConvTranspose2d)
    w = consumer.weight
    if is_t:
        ok = SPADE_Y16 and L.norm_output_as_bf16(N, Cc, H, W, "convT", w.shape[1], need_bww=w.requires_grad)
        kind, stride, pad, bias = "convT", 2, 1, None
    else:
        ks, stride, pad, bias = consumer.kernel_size[0], consumer.stride[0], consumer.padding[0], consumer.bias
        # the modulation applied by the consumer itself: where gamma|beta's class grid is small against the map (its cell table is
        # read in place of a second copy of the activation) and the row pass of the backward reduces to that grid (64 / 128 wide maps)
        if (SPADE_FOLD and g is not None and W in (64, 128) and H == W and 4 * gb.shape[-1] ** 2 <= H * W
                and L.conv_spade_ok(N, Cc, H, W, w.shape[0], ks, stride, pad, need_bww=w.requires_grad)):
            return _SpadeFoldConv.apply(x, gb, rmean, rvar, nbt, relu, training, g, w, bias, stride, pad)
        ok = SPADE_Y16 and L.norm_output_as_bf16(N, Cc, H, W, "conv", w.shape[0], ks, stride, pad, need_bww=w.requires_grad)
        kind = "conv"
    if ok:
        return _SpadeThenConv.apply(x, gb, rmean, rvar, nbt, relu, training, g, w, bias, kind, stride, pad)
    return consumer(_NormAct.apply(x, gb, None, None, None, rmean, rvar, nbt, 3, relu, training, g))


class _NormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p0, p1, residual, labels, rmean, rvar, nbt, mode, relu, training, gather=None):
        ctx.pslots = (_slot(p0), _slot(p1))      # arena gradient slots of the parameters (affine gamma/beta; class table)
        # gather (mode 3): (map, range starts of the inverse map) of _grid_map — p0 lives on the coarser block-class grid
        ctx.gather = gather
        gmap = gather[0] if gather is not None else None
        x = _c(x)
        mean, rstd = _batch_statistics(x, rmean, rvar, nbt, training)
        p0c = _c(p0) if p0 is not None else None
        y = L.norm_apply_fwd(x, mean, rstd, mode, p0c, p1, labels, _c(residual) if residual is not None else None, relu, gb_map=gmap)
        ctx.cfg = (mode, relu, training, residual is not None)
        ctx.save_for_backward(x, y if relu else None, mean, rstd, p0c, p1, labels)
        return y

    @staticmethod
    def backward(ctx, dy):
        mode, relu, training, has_res = ctx.cfg
        x, y, mean, rstd, p0, p1, labels = ctx.saved_tensors
        dy = _c(dy)
        dp0 = dp1 = None
        s0, s1 = ctx.pslots
        in_slots = False          # parameter gradients accumulated by the kernel straight into the arena (no autograd add launch)
        if mode == 1:
            if s0 is not None and s1 is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
                dp0, dp1, in_slots = s0, s1, True
            else:
                dp0, dp1 = torch.empty_like(p0), torch.empty_like(p1)
        elif mode == 2:
            if s0 is not None and ctx.needs_input_grad[1]:
                dp0, in_slots = s0, True
            else:
                dp0 = torch.zeros_like(p0)
        gather = ctx.gather
        # gathered gamma|beta on 64- / 128-wide maps: the row pass reduces d(gamma|beta) to the class grid itself
        fused_reduce = gather is not None and mode == 3 and x.shape[-1] in (64, 128) and x.shape[-2] == x.shape[-1]
        if mode == 3:        # d(gamma|beta): on the class grid when reduced in the kernel, else at the map's full resolution
            dp0 = (torch.empty_like(p0) if fused_reduce else
                   torch.empty((x.shape[0], 2 * x.shape[1]) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device))
        dx = L.norm_bwd(dy, x, y, mean, rstd, mode, p0, p1, labels, relu, training, dp0, dp1, param_accumulate=in_slots,
                        gb_map=gather[0] if gather is not None else None, gb_lo=gather[1] if fused_reduce else None)
        if in_slots:
            dp0 = dp1 = None
        if gather is not None and not fused_reduce:
            lo, h = gather[1], p0.shape[-1]
            dsrc = torch.empty_like(p0)
            L.call("agl_grid_gather_bwd", L.ptr(dp0), L.ptr(lo, torch.int32), L.ptr(lo, torch.int32), L.ptr(dsrc), dp0.shape[0] * dp0.shape[1],
                   h, h, dp0.shape[-1], dp0.shape[-1], L.stream())
            dp0 = dsrc
        dres = None
        if has_res and ctx.needs_input_grad[3]:
            dres = L.relu_bwd(dy, y) if relu else dy
        return dx, dp0, dp1, dres, None, None, None, None, None, None, None, None


def batch_norm(x, rmean, rvar, nbt, weight=None, bias=None, relu=False, residual=None, training=True):
    """nn.BatchNorm2d / nn.BatchNorm1d (+ReLU, +residual add)."""
    mode = 1 if weight is not None else 0
    return _NormAct.apply(x, weight, bias, residual, None, rmean, rvar, nbt, mode, relu, training)


def cond_batch_norm(x, table, labels, rmean, rvar, nbt, relu=False, training=True):
    """ConditionalBatchNorm2d (generator_obj_att.py:31-44): gamma|beta rows of `table` picked by labels."""
    return _NormAct.apply(x, table, None, None, labels, rmean, rvar, nbt, 2, relu, training)


def spade_modulate(x, gb, rmean, rvar, nbt, relu=False, training=True, gather=None):
    """SPADE apply (normalization.py:97,106): y = BN(x) * (1 + gamma) + beta with gb = [gamma; beta].
    gather = (kind, blocks, f) of _grid_map: gb lives on that block-class grid and is expanded while it is read (what
    grid_gather(gb, kind, blocks, f) would write out first)."""
    g = None
    if gather is not None:
        m, lo, src = _grid_map(gather[0], gather[1], gather[2], x.device)
        assert gb.shape[2] == gb.shape[3] == src and x.shape[2] == x.shape[3] == m.numel(), (gb.shape, x.shape, gather)
        g = (m, lo)
    return _NormAct.apply(x, gb, None, None, None, rmean, rvar, nbt, 3, relu, training, g)


# --------------------------------------------------------------------------- crop
class _Crop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, boxes, o2i, HH, WW, align):
        feats, boxes = _c(feats), _c(boxes)
        ctx.cfg = (tuple(feats.shape), align, getattr(o2i, "_agl_sorted", False))
        ctx.save_for_backward(boxes, o2i)
        return L.crop_fwd(feats, boxes, o2i, HH, WW, align)

    @staticmethod
    def backward(ctx, dout):
        shape, align, in_order = ctx.cfg
        boxes, o2i = ctx.saved_tensors
        if in_order:      # (saved tensors may come back as new Python objects: the mark travels in ctx)
            o2i._agl_sorted = True
        return L.crop_bwd(_c(dout), boxes, o2i, shape, align), None, None, None, None, None


def crop_boxes(feats, boxes, box_to_img_dev, HH, WW=None, align_corners=False):
    return _Crop.apply(feats, boxes, box_to_img_dev, HH, HH if WW is None else WW, align_corners)


# --------------------------------------------------------------------------- pooling / resampling
class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, in_relu):
        x = _c(x)
        ctx.in_relu = in_relu
        ctx.shape = tuple(x.shape)
        ctx.save_for_backward(x if in_relu else None)
        return L.avgpool2_fwd(x, in_relu)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return L.avgpool2_bwd(_c(dy), x if ctx.in_relu else ctx.shape, ctx.in_relu), None


def avg_pool2(x, in_relu=False):
    return _AvgPool2.apply(x, in_relu)


class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        return L.upsample_fwd(_c(x), k)

    @staticmethod
    def backward(ctx, dy):
        return L.upsample_bwd(_c(dy), ctx.k), None


def upsample_nearest(x, log2_factor):
    return _Upsample.apply(x, log2_factor)


class _SumHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, in_relu, scale):
        x = _c(x)
        ctx.cfg = (in_relu, scale)
        ctx.save_for_backward(x)
        return L.sum_hw_fwd(x, in_relu, scale)

    @staticmethod
    def backward(ctx, dy):
        in_relu, scale = ctx.cfg
        (x,) = ctx.saved_tensors
        return L.sum_hw_bwd(_c(dy), x, in_relu, scale), None, None


def sum_hw(x, in_relu=False, scale=1.0):
    """(N,C,H,W) -> (N,C): scale * sum over H,W of (relu)(x)."""
    return _SumHW.apply(x, in_relu, scale)


class _Reparam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _c(mu), _c(logvar), _c(eps)
        ctx.save_for_backward(logvar, eps)
        z = torch.empty_like(mu)
        L.call("agl_reparam_fwd", L.ptr(mu), L.ptr(logvar), L.ptr(eps), L.ptr(z), mu.numel(), L.stream())
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = _c(dz)
        dlv = torch.empty_like(dz)
        L.call("agl_reparam_bwd", L.ptr(dz), L.ptr(logvar), L.ptr(eps), L.ptr(dlv), dz.numel(), L.stream())
        return dz, dlv, None


def reparameterize(mu, logvar, eps):
    return _Reparam.apply(mu, logvar, eps)


class _MaskOuter(torch.autograd.Function):
    """u (O,C) x zero-padded mask (O,1,R,R) -> (O,C,R+2p,R+2p): LayoutEncoder's rank-1 input pushed
    through its 1x1/pad-1 convolution without building the (O,128,R,R) tensor."""

    @staticmethod
    def forward(ctx, u, mask, pad):
        u, mask = _c(u), _c(mask)
        O, Cc = u.shape
        R = mask.shape[-1]
        ctx.pad = pad
        ctx.save_for_backward(mask)
        y = torch.empty((O, Cc, R + 2 * pad, R + 2 * pad), dtype=torch.float32, device=u.device)
        L.call("agl_mask_outer_fwd", L.ptr(u), L.ptr(mask), L.ptr(y), O, Cc, R, pad, L.stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _c(dy)
        O, Cc = dy.shape[0], dy.shape[1]
        du = torch.empty((O, Cc), dtype=torch.float32, device=dy.device)
        L.call("agl_mask_outer_bwd", L.ptr(dy), L.ptr(mask), L.ptr(du), O, Cc, mask.shape[-1], ctx.pad, L.stream())
        return du, None, None


def mask_outer(u, mask, pad=1):
    return _MaskOuter.apply(u, mask, pad)


class _PoolFuseWeight(torch.autograd.Function):
    """(Cout,Cin,3,3) -> (Cout,Cin,4,4): the 4x4/stride-2 filter equal to conv3x3(pad 1) followed by avg_pool2d(2)."""

    @staticmethod
    def forward(ctx, w3):
        w3 = _c(w3)
        assert w3.shape[2:] == (3, 3)
        w4 = torch.empty(w3.shape[:2] + (4, 4), dtype=torch.float32, device=w3.device)
        L.call("agl_pool_fuse_weight_fwd", L.ptr(w3), L.ptr(w4), w3.shape[0] * w3.shape[1], L.stream())
        return w4

    @staticmethod
    def backward(ctx, dw4):
        dw4 = _c(dw4)
        dw3 = torch.empty(dw4.shape[:2] + (3, 3), dtype=torch.float32, device=dw4.device)
        L.call("agl_pool_fuse_weight_bwd", L.ptr(dw4), L.ptr(dw3), dw4.shape[0] * dw4.shape[1], L.stream())
        return dw3


class _Box2(torch.autograd.Function):
    """2x2 stride-1 box filter over the zero-extended map ((H+1) x (W+1) outputs); mask_grad: the input is a ReLU
    output whose producer skipped its ReLU-backward pass, so the returned gradient is masked by x > 0."""

    @staticmethod
    def forward(ctx, x, mask_grad):
        x = _c(x)
        ctx.save_for_backward(x if mask_grad else None)
        return L.box2_fwd(x)

    @staticmethod
    def backward(ctx, dxb):
        (x,) = ctx.saved_tensors
        return L.box2_bwd(_c(dxb), x), None


_grid_maps = {}


def _grid_map(kind: str, blocks: int, f: int, device):
    """Index maps between the grids of an image made of `blocks` constant f x f blocks per axis (a nearest up-sampled
    map) and of 3x3 convolutions of it, cached on the device as (map, range starts of the inverse):
      'up3'   8  -> 3 per block : every block three times (a 3x3 convolution of this grid takes, at offsets 0/1/2 of a
              block, exactly the values it takes at the first / an interior / the last row of the f-fold up-sampling);
      '3to5'  3 per block -> 5 per block: classes (first, second, interior, last but one, last) read (0, 1, 1, 1, 2);
      '5tof'  5 per block -> f per block: rows 0, 1, 2..f-3, f-2, f-1 read classes 0, 1, 2, 3, 4   (f >= 5);
      '3tof'  3 per block -> f per block: rows 0, 1..f-2, f-1 read classes 0, 1, 2                 (f >= 3)."""
    key = (kind, blocks, f, str(device))
    if key not in _grid_maps:
        if kind == "up3":
            src, per, pat = blocks, 3, [0, 0, 0]
            m = [b for b in range(blocks) for _ in range(3)]
        else:
            pat = {"3to5": [0, 1, 1, 1, 2], "5tof": [0, 1] + [2] * (f - 4) + [3, 4], "3tof": [0] + [1] * (f - 2) + [2]}[kind]
            spb = {"3to5": 3, "5tof": 5, "3tof": 3}[kind]            # source cells per block
            src = blocks * spb
            m = [b * spb + c for b in range(blocks) for c in pat]
        lo = [0] * (src + 1)
        for v in m:
            lo[v + 1] += 1
        for i in range(src):
            lo[i + 1] += lo[i]
        _grid_maps[key] = (torch.tensor(m, dtype=torch.int32, device=device), torch.tensor(lo, dtype=torch.int32, device=device), src)
    return _grid_maps[key]


class _GridGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m, lo):
        x = _c(x)
        N, Cc, h, w = x.shape
        H = m.numel()
        assert h == w == lo.numel() - 1
        y = torch.empty((N, Cc, H, H), dtype=torch.float32, device=x.device)
        L.call("agl_grid_gather_fwd", L.ptr(x), L.ptr(m, torch.int32), L.ptr(m, torch.int32), L.ptr(y), N * Cc, h, w, H, H, L.stream())
        ctx.save_for_backward(lo)
        ctx.shape = (N, Cc, h, w, H)
        return y

    @staticmethod
    def backward(ctx, dy):
        (lo,) = ctx.saved_tensors
        N, Cc, h, w, H = ctx.shape
        dx = torch.empty((N, Cc, h, w), dtype=torch.float32, device=dy.device)
        L.call("agl_grid_gather_bwd", L.ptr(_c(dy)), L.ptr(lo, torch.int32), L.ptr(lo, torch.int32), L.ptr(dx), N * Cc, h, w, H, H, L.stream())
        return dx, None, None


def grid_gather(x, kind: str, blocks: int, f: int = 0):
    """Square-map index gather along both axes with one of the cached maps of _grid_map."""
    m, lo, src = _grid_map(kind, blocks, f, x.device)
    assert x.shape[2] == x.shape[3] == src, (x.shape, kind, blocks, f)
    return _GridGather.apply(x, m, lo)


class _Conv3x3AvgPool(torch.autograd.Function):
    """avg_pool2d(conv2d(x, w3, bias, padding=1), 2) with the cheapest exact form per pass:
      forward         3x3 stride-2 convolution of the box-filtered zero-extended input xb      (9 taps per output)
      weight gradient 3x3 stride-2 weight gradient against xb                                  (9 taps)
      input gradient  maps >= 16 wide: 3x3 stride-2 input gradient w.r.t. xb (stride phases of 4:2:2:1 taps, 9 in total)
                      followed by the transposed box filter; smaller maps: 4x4 stride-2 input gradient with the pooled
                      filter w4 (four equally sized phases of 2x2 taps — the unequal phases measured slower there);
                      either way with the ReLU mask of a producer that left its ReLU backward to this consumer."""

    @staticmethod
    def forward(ctx, x, w3, bias, x_relu):
        ctx.slots = (_slot(w3), _slot(bias))
        ctx.wsrc = getattr(w3, "_agl_wsrc", None)
        x, w3 = _c(x), _c(w3)
        # bf16 arithmetic: the filtered map is read by this convolution (and its weight gradient) only — written as bf16 it holds the
        # very values they would round it to, at half the traffic (where both run on the matrix-core kernels; BOX_BF16 False: fp32)
        N, Cc, H, W = x.shape
        as_bf16 = BOX_BF16 and L.box_input_as_bf16(N, Cc, H + 1, W + 1, w3.shape[0], need_bww=ctx.needs_input_grad[1])
        xb = L.box2_fwd(x, bf16=as_bf16)
        y = L.conv2d_fwd(xb, w3, bias, 2, 0, wsrc=ctx.wsrc)
        ctx.cfg = (x_relu, tuple(x.shape), bias is not None)
        ctx.save_for_backward(x if x_relu else None, xb, w3)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_relu, xshape, has_bias = ctx.cfg
        x, xb, w3 = ctx.saved_tensors
        dy = _c(dy)
        dx = dw = db = None
        # bf16-matrix-core modes: the 4x4 / stride-2 form always (its four even phases run on csrc/pconv.hip; the 3x3 / stride-2
        # phases of the box form are unequal and stay on the im2col kernel)
        mc = bool(L.CONV_FLAGS & (L.CONV_BF16 | L.CONV_SPLIT3))
        if ctx.needs_input_grad[0] and not mc and (BOX_BWD or xshape[2] >= BOX_BWD_MIN):
            dxb = L.conv2d_bwd_data(dy, w3, (xshape[2] + 1, xshape[3] + 1), 2, 0)
            dx = L.box2_bwd(dxb, x if x_relu else None)
        elif ctx.needs_input_grad[0]:
            def pooled(src):
                w4 = torch.empty(src.shape[:2] + (4, 4), dtype=torch.float32, device=src.device)
                L.call("agl_pool_fuse_weight_fwd", L.ptr(src), L.ptr(w4), src.shape[0] * src.shape[1], L.stream())
                return w4
            ws = ctx.wsrc         # the pooled filter is only built when its packed form is not cached (once per weight version)
            dx = L.conv2d_bwd_data(dy, None, (xshape[2], xshape[3]), 2, 1, pos_mask=x if x_relu else None,
                                   wsrc=ws.derived("pool4") if ws is not None else None, w_shape=tuple(w3.shape[:2]) + (4, 4),
                                   make_w=lambda: pooled(w3),
                                   make_base=lambda: pooled(_c(ws.base.detach()) if ws.base is not None else w3))
        wslot, bslot = ctx.slots
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if wslot is not None:
                fuse_b = want_b and bslot is not None
                L.on_wgrad_stream(lambda: L.conv2d_bwd_weight(dy, xb, 3, 2, 0, out=wslot, accumulate=True, dbias=bslot if fuse_b else None),
                                  dy, xb)
            else:
                fuse_b = want_b
                if fuse_b and bslot is None:
                    db = torch.empty(w3.shape[0], dtype=torch.float32, device=dy.device)
                dw = L.conv2d_bwd_weight(dy, xb, 3, 2, 0, dbias=(bslot if bslot is not None else db) if fuse_b else None,
                                         dbias_accumulate=bslot is not None)
            want_b = want_b and not fuse_b
        if want_b:
            if bslot is not None:
                L.channel_sum(dy, out=bslot, accumulate=True)
            else:
                db = L.channel_sum(dy)
        return dx, dw, db, None


# (Closed A/B experiments of rounds 2-4: plain module constants now — tests flip them in process; no environment switches.)
BOX_BF16 = True       # bf16 arithmetic stores the box-filtered maps as bf16 (False: fp32)
H_BF16 = True         # ... and the first-convolution output of a flat discriminator block
BOX_FORM = True       # False: the 4x4 stride-2 form with the pooled filter (equivalence tests)
BOX_BWD = False       # input gradient through the 3x3/stride-2 phases + box transpose
BOX_BWD_MIN = 16      # ... used from this map size up (measured: 0.24 vs 0.34 ms at
                                                             # 32x32 and 16x16, but 0.48 vs 0.30 ms at 8x8)


def conv3x3_avgpool2(x, w3, bias=None, in_relu=False, x_relu=False):
    """avg_pool2d(conv2d(x, w3, bias, padding=1), 2), exact in real arithmetic, in one of two fused forms:
      * box form: avgpool o conv3x3 = (3x3 stride-2 unpadded convolution) o (2x2 stride-1 box filter of the
        zero-extended input) — 9 taps per output (4x fewer MACs than conv-then-pool), one cheap elementwise pass;
      * pooled-filter form: one 4x4 stride-2 convolution with w4 = 1/4 sum of the four shifted copies of w3 (16 taps)."""
    if BOX_FORM and not in_relu and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
        return _Conv3x3AvgPool.apply(x, w3, bias, x_relu)
    return conv2d(x, _PoolFuseWeight.apply(w3), bias, 2, 1, 0, in_relu, False, None, False, x_relu)


class _Concat2(torch.autograd.Function):
    """cat((a, b), dim=1) for (N,Ca,*spatial) and (N,Cb,*spatial) — or b of shape (N,Cb) broadcast over the spatial dims."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        N, Ca, Cb = a.shape[0], a.shape[1], b.shape[1]
        HW = a.numel() // (N * Ca)
        bcast = int(b.dim() == 2 and a.dim() > 2)
        assert bcast or b.numel() == N * Cb * HW
        out = torch.empty((N, Ca + Cb) + tuple(a.shape[2:]), dtype=torch.float32, device=a.device)
        L.call("agl_concat2_fwd", L.ptr(a), L.ptr(b), L.ptr(out), N, Ca, Cb, HW, bcast, L.stream())
        ctx.cfg = (tuple(a.shape), tuple(b.shape), HW, bcast)
        return out

    @staticmethod
    def backward(ctx, d):
        sa, sb, HW, bcast = ctx.cfg
        d = _c(d)
        da = torch.empty(sa, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(sb, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            L.call("agl_concat2_bwd", L.ptr(d), L.ptr(da), L.ptr(db), sa[0], sa[1], sb[1], HW, bcast, L.stream())
        return da, db


def concat_channels(a, b):
    return _Concat2.apply(a, b)


def concat_rows(a, b):
    """cat((a, b), dim=0) of two contiguous tensors with equal trailing shape (a flat copy)."""
    y = _Concat2.apply(a.reshape(1, -1), b.reshape(1, -1))
    return y.reshape((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]))


class _Split2(torch.autograd.Function):
    """x (N, Ca+Cb, *rest) -> contiguous copies x[:, :Ca], x[:, Ca:] (the adjoint of concat_channels)."""

    @staticmethod
    def forward(ctx, x, Ca):
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        a = torch.empty((N, Ca) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        b = torch.empty((N, Cc - Ca) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        L.call("agl_concat2_bwd", L.ptr(x), L.ptr(a), L.ptr(b), N, Ca, Cc - Ca, HW, 0, L.stream())
        ctx.cfg = (Ca, Cc - Ca, HW)
        return a, b

    @staticmethod
    def backward(ctx, da, db):
        Ca, Cb, HW = ctx.cfg
        da, db = _c(da), _c(db)
        out = torch.empty((da.shape[0], Ca + Cb) + tuple(da.shape[2:]), dtype=torch.float32, device=da.device)
        L.call("agl_concat2_fwd", L.ptr(da), L.ptr(db), L.ptr(out), da.shape[0], Ca, Cb, HW, 0, L.stream())
        return out, None


def split_channels(x, Ca):
    return _Split2.apply(x, Ca)


class _LayoutStage1(torch.autograd.Function):
    """c2(relu(CondBN(c0-output))) of the layout encoder for the rank-1 input u (x) mask, in closed form
    (csrc/layout.hip): the (O,64,R+2,R+2) activations are never built and c2 costs 32 FMAs per output."""

    @staticmethod
    def forward(ctx, u, masks, labels, table, w2, rmean, rvar, nbt, training):
        ctx.tslot = _slot(table)
        u, masks, table, w2 = _c(u), _c(masks), _c(table), _c(w2)
        O, Cc = u.shape
        R, Co = masks.shape[-1], w2.shape[0]
        assert w2.shape[1:] == (Cc, 4, 4) and masks.shape[-2] == R
        dev = u.device
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        area, A, B, D = f(O), f(O, Cc), f(O, Cc), f(O, Cc)
        if training:
            mean, rstd = f(Cc), f(Cc)
        else:
            mean, rstd = L.bn_stats_eval(rmean, rvar, BN_EPS)
        def levels(area, mean, rstd, A, B, D):
            L.call("agl_layout1_levels", L.ptr(u), L.ptr(masks), L.ptr(labels, torch.int64), L.ptr(table), L.ptr(area), L.ptr(mean),
                   L.ptr(rstd), L.ptr(A), L.ptr(B), L.ptr(D), L.ptr(rmean), L.ptr(rvar), L.ptr(nbt, torch.int64), O, Cc, R, BN_EPS,
                   BN_MOMENTUM, int(training), L.stream())
        levels(area, mean, rstd, A, B, D)
        if training and BN_TAPE is not None and rmean is not None:      # replay into scratch outputs: only the running stats move
            BN_TAPE.append(lambda: levels(f(O), f(Cc), f(Cc), f(O, Cc), f(O, Cc), f(O, Cc)))
        Wr = f(Co * 16, Cc, 1, 1)
        L.call("agl_layout1_permute", L.ptr(w2), L.ptr(Wr), Co, Cc, 1, L.stream())
        WB = L.conv2d_fwd(B.view(O, Cc, 1, 1), Wr)
        WD = L.conv2d_fwd(D.view(O, Cc, 1, 1), Wr)
        OH = R // 2 + 1
        y = f(O, Co, OH, OH)
        L.call("agl_layout1_pixels", L.ptr(WB), L.ptr(WD), L.ptr(masks), L.ptr(y), O, Co, R, L.stream())
        ctx.cfg = (training, R, Co)
        ctx.save_for_backward(u, masks, labels, table, area, mean, rstd, A, B, D, Wr)
        return y

    @staticmethod
    def backward(ctx, dy):
        training, R, Co = ctx.cfg
        u, masks, labels, table, area, mean, rstd, A, B, D, Wr = ctx.saved_tensors
        dy = _c(dy)
        O, Cc = u.shape
        dev = u.device
        GB = torch.empty((O, Co * 16, 1, 1), dtype=torch.float32, device=dev)
        GD = torch.empty_like(GB)
        L.call("agl_layout1_tapsum", L.ptr(dy), L.ptr(masks), L.ptr(GB), L.ptr(GD), O, Co, R, L.stream())
        dW2 = None
        if ctx.needs_input_grad[4]:
            dWr = L.axpby(L.conv2d_bwd_weight(GB, B.view(O, Cc, 1, 1), 1), L.conv2d_bwd_weight(GD, D.view(O, Cc, 1, 1), 1))
            dW2 = torch.empty((Co, Cc, 4, 4), dtype=torch.float32, device=dev)
            L.call("agl_layout1_permute", L.ptr(dWr), L.ptr(dW2), Co, Cc, 0, L.stream())
        dA = L.conv2d_bwd_data(GD, Wr, (1, 1)).view(O, Cc)
        dB = L.conv2d_bwd_data(L.axpby(GB, GD, 1.0, -1.0), Wr, (1, 1)).view(O, Cc)
        du = torch.empty_like(u)
        tslot = ctx.tslot if ctx.needs_input_grad[3] else None      # the kernel adds into dtable: straight into the arena slot
        dtable = tslot if tslot is not None else (torch.zeros_like(table) if ctx.needs_input_grad[3] else None)
        ws = L.workspace((O * Cc * 4 + 2 * Cc) * 4, dev)
        L.call("agl_layout1_levels_bwd", L.ptr(dA), L.ptr(dB), L.ptr(A), L.ptr(B), L.ptr(u), L.ptr(area), L.ptr(mean), L.ptr(rstd),
               L.ptr(table), L.ptr(labels, torch.int64), L.ptr(du), L.ptr(dtable), O, Cc, R, table.shape[0], int(training),
               ws.data_ptr(), ws.numel(), L.stream())
        return du, None, None, (None if tslot is not None else dtable), dW2, None, None, None, None


def layout_stage1(u, masks, labels, table, w2, rmean, rvar, nbt, training=True):
    return _LayoutStage1.apply(u, masks, labels, table, w2, rmean, rvar, nbt, training)


class _CatBatch(torch.autograd.Function):
    """Concatenate equally shaped tensors along dim 0 (plumbing copies; backward hands out views)."""

    @staticmethod
    def forward(ctx, *xs):
        n = xs[0].shape[0]
        out = torch.empty((n * len(xs),) + tuple(xs[0].shape[1:]), dtype=xs[0].dtype, device=xs[0].device)
        for i, x in enumerate(xs):
            assert x.shape == xs[0].shape
            out[i * n:(i + 1) * n].copy_(x)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, d):
        n = ctx.n
        return tuple(d[i * n:(i + 1) * n] for i in range(d.shape[0] // n))


class _SplitBatch(torch.autograd.Function):
    """Inverse of _CatBatch: k equal chunks along dim 0 as views; backward gathers the chunk gradients."""

    @staticmethod
    def forward(ctx, x, k):
        n = x.shape[0] // k
        ctx.shape = tuple(x.shape)
        outs = tuple(x[i * n:(i + 1) * n] for i in range(k))
        return outs

    @staticmethod
    def backward(ctx, *ds):
        out = torch.empty(ctx.shape, dtype=torch.float32, device=next(d for d in ds if d is not None).device)
        n = ctx.shape[0] // len(ds)
        for i, d in enumerate(ds):
            if d is None:
                out[i * n:(i + 1) * n].zero_()
            else:
                out[i * n:(i + 1) * n].copy_(d)
        return out, None


def cat_batch(xs):
    return _CatBatch.apply(*xs)


def split_batch(x, k):
    return _SplitBatch.apply(x, k)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return L.axpby(_c(a), _c(b), 1.0, 1.0)

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return _Add.apply(a, b)


class _GatherRows(torch.autograd.Function):
    """Row gather (nn.Embedding lookup); backward is a fixed-order row accumulation."""

    @staticmethod
    def forward(ctx, table, rows):
        table = _c(table)
        ctx.n = table.shape[0]
        ctx.save_for_backward(rows)
        return L.gather_rows(table, rows)

    @staticmethod
    def backward(ctx, dout):
        (rows,) = ctx.saved_tensors
        dout = _c(dout)
        D = dout.numel() // dout.shape[0]
        dt = torch.zeros((ctx.n,) + tuple(dout.shape[1:]), dtype=dout.dtype, device=dout.device)
        L.call("agl_embedding_bwd", L.ptr(dout), L.ptr(rows, torch.int64), L.ptr(dt), dout.shape[0], D, ctx.n, L.stream())
        return dt, None


def embedding(table, rows):
    return _GatherRows.apply(table, rows)


# --------------------------------------------------------------------------- spectral norm
class _SpectralNormWeights(torch.autograd.Function):
    """All spectrally-normalised weights of one discriminator for ONE forward call
    (torch.nn.utils.spectral_norm semantics: one power iteration per training forward)."""

    @staticmethod
    def forward(ctx, training, us, vs, *ws):
        dev = ws[0].device
        n = len(ws)
        sizes = [w.numel() for w in ws]
        rows = [w.shape[0] for w in ws]
        cols = [s // r for s, r in zip(sizes, rows)]
        arena = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        sigma = torch.empty(n, dtype=torch.float32, device=dev)
        u_used = torch.empty(sum(rows), dtype=torch.float32, device=dev)
        v_used = torch.empty(sum(cols), dtype=torch.float32, device=dev)
        lib = L.load()
        tmp_off, tot = [], 0
        for r, c in zip(rows, cols):                      # every layer gets its own scratch slice
            tmp_off.append(tot)
            tot += lib.agl_sn_tmp_floats(r, c)
        tmp = torch.empty(tot, dtype=torch.float32, device=dev)
        descs = (L.SnLayer * n)()
        outs, held, o, ro, co = [], [], 0, 0, 0
        for i, w in enumerate(ws):
            wc = _c(w)
            held.append(wc)
            d = descs[i]
            d.w, d.u, d.v = L.ptr(wc), L.ptr(us[i]), L.ptr(vs[i])
            d.w_sn = arena.data_ptr() + 4 * o
            d.sigma = sigma.data_ptr() + 4 * i
            d.tmp = tmp.data_ptr() + 4 * tmp_off[i]
            d.u_used = u_used.data_ptr() + 4 * ro
            d.v_used = v_used.data_ptr() + 4 * co
            d.g, d.dw = None, None
            d.rows, d.cols = rows[i], cols[i]
            outs.append(arena[o:o + sizes[i]].view(w.shape))
            o, ro, co = o + sizes[i], ro + rows[i], co + cols[i]
        L.call("agl_sn_forward", C.cast(descs, C.c_void_p), n, int(training), SN_EPS, L.stream())
        ctx.descs, ctx.keep = descs, (arena, sigma, u_used, v_used, tmp)
        ctx.shapes = [tuple(w.shape) for w in ws]
        slots = [_slot(w) for w in ws]
        ctx.slots = slots if all(sl is not None for sl in slots) else None
        ctx.mark_non_differentiable(sigma)
        return tuple(outs) + (sigma,)

    @staticmethod
    def backward(ctx, *gs):
        gs = gs[:-1]                 # (the last output is sigma: non-differentiable)
        descs = ctx.descs
        arena = ctx.keep[0]
        n = len(gs)
        slots = ctx.slots
        dws = torch.empty_like(arena) if slots is None else None
        held, o = [], 0
        for i, g in enumerate(gs):
            if g is None:
                g = torch.zeros(ctx.shapes[i], dtype=torch.float32, device=arena.device)
            g = _c(g)
            held.append(g)
            descs[i].g = g.data_ptr()
            descs[i].dw = slots[i].data_ptr() if slots is not None else dws.data_ptr() + 4 * o
            o += g.numel()
        L.call("agl_sn_backward", C.cast(descs, C.c_void_p), n, int(slots is not None), L.stream())
        if slots is not None:                      # accumulated straight into the weight_orig gradient slots
            return (None, None, None) + (None,) * n
        outs, o = [], 0
        for i in range(n):
            sz = held[i].numel()
            outs.append(dws[o:o + sz].view(ctx.shapes[i]))
            o += sz
        return (None, None, None) + tuple(outs)


def spectral_norm_weights(weights: Sequence[torch.Tensor], us: Sequence[torch.Tensor], vs: Sequence[torch.Tensor],
                          training: bool = True) -> List[torch.Tensor]:
    outs = _SpectralNormWeights.apply(training, list(us), list(vs), *weights)
    ws, sigma = list(outs[:-1]), outs[-1]
    # packed-weight cache: the convolutions read the packed weight_orig (re-packed once per optimiser update) and divide by
    # this call's sigma in their epilogue
    for i, (w_sn, w) in enumerate(zip(ws, weights)):
        src = getattr(w, "_agl_wsrc", None)
        if src is not None:
            w_sn._agl_wsrc = L.WeightSrc(src.owner, src.version, base=w, div=sigma[i:i + 1], tag="sn")
    return ws
