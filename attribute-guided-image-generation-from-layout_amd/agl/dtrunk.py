"""The block chain of a discriminator (reference models/discriminator.py:29-60 OptimizedBlock, :63-99 ResidualBlock, chained at
:194-201, :243-250, :112-120) as ONE autograd node in bf16 arithmetic.

Inside the node no tensor is an autograd edge, so every activation whose only readers are convolutions (and the 2x2 average pool
of the shortcut) is stored as bf16 — half the bytes on the tensors that bound the discriminators at 128 px:

  first block without down-sampling (object / attribute discriminator, x = a 3-channel crop):
      h   = relu(c1(x))                        few-input-channel stream kernel, written as bf16
      out = c2(h) + b2 + sc(x)                 ONE 3x3 launch: the 1x1 shortcut of the 3-channel x is evaluated in its epilogue
                                               (agl_conv2d_fwd_shortcut), the sum is rounded once and written as bf16
  down-sampling block (input o: fp32 or bf16):
      h   = relu(c1(relu(o)))                  3x3, written as bf16 by the convolution's epilogue
      s   = avg_pool2(relu(o))                 (the block's in-place ReLU aliases the shortcut input, discriminator.py:71,99)
      hp  = avg_pool2(c2(h))                   ONE 4x4 / stride-2 convolution of h with the pooled filter (w4 = the four shifted
                                               copies of w3, packed once per weight version) — no box-filtered copy of h
      out = hp + sc(s)                         1x1 convolution with hp as out-of-place addend; bf16 when another covered block follows

What a reader rounds anyway is what gets stored: the convolutions stage bf16 operands, so c1 / c2 / sc see the values they would
have computed from fp32 tensors; the pooled shortcut reads the rounded o (2^-9 relative per element before the mean of four).
Backward is the hand-composed chain of the same launches the per-op graph issues (input gradients with the ReLU masks read from the
bf16 tensors, weight gradients with bf16 x operands); the pooled-filter weight gradient is mapped back to the 3x3 filter.

`cover()` decides per call which prefix of blocks runs here: every launch of a covered block must be one of the matrix-core forms
that read / write bf16 (asked through the C ABI's predicates); the remaining blocks run as before.  Only in bf16 arithmetic.

CHANNEL-BLOCKED FORM (D_BLOCKED, round 5).  The bf16 tensors of the node — h of every block and the block outputs handed to the next
covered block — are stored as [N][C/8][H][W][8] (include/agl.h AGL_CONV_X_BLOCKED ...): the convolutions that read them stage one aligned
16-byte piece per 8 channels of a pixel instead of eight 2-byte element loads a channel stride apart (the staging pass of the NCHW form is
what bounds the bf16-mode kernels), and the producers store half a piece per lane straight from the accumulators.  Readers:
c1 (3x3, blocked in -> blocked out), the pooled 4x4 / stride-2 convolution (blocked in -> fp32 NCHW out), both weight gradients (blocked x),
the input gradients (blocked ReLU mask; dy / dx stay fp32 NCHW), the shortcut's average pool and its backward (blocked in / blocked mask).
An fp32 block input (the image discriminator's first covered block) is converted once (agl_to_blocked) for the convolution operands; its
pool and masks keep reading the fp32 tensor.  Same values as the NCHW bf16 form in every element (tests: bit identity of the node's output
and of every gradient against D_BLOCKED = False); taken when every launch of every covered block has its blocked form (blocked_ok).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import functional as F
from . import lib as L

D_TRUNK = True       # False (tests flip it in process) = the per-op graph in every arithmetic
D_BLOCKED = True     # False (tests flip it in process) = NCHW bf16 tensors inside the node (the round-4 form)


def _pooled(w3):
    w4 = torch.empty(tuple(w3.shape[:2]) + (4, 4), dtype=torch.float32, device=w3.device)
    L.call("agl_pool_fuse_weight_fwd", L.ptr(w3), L.ptr(w4), w3.shape[0] * w3.shape[1], L.stream())
    return w4


def _pooled_src(w, ws):
    """Tensor whose pooled 4x4 form is packed under a WeightSrc (the un-normalised weight of a spectrally normalised layer)."""
    return _pooled(F._c(ws.base.detach()) if (ws is not None and ws.base is not None) else w)


def _flags_ok():
    return bool(L.CONV_FLAGS & L.CONV_BF16) and not (L.CONV_FLAGS & L.CONV_NO_PATCH)


def flat_first_ok(N, H, W, C):
    """First block without down-sampling on an (N, 3, H, W) input with C channels inside."""
    lib, fl = L.load(), L.CONV_FLAGS
    return (W % 8 == 0 and L.first_conv_output_as_bf16(N, 3, H, W, C, C, 3, need_bww=True, need_bwd_data=True)
            and bool(lib.agl_conv2d_fwd_shortcut_ok(N, C, H, W, C, 3, 1, fl))
            and bool(lib.agl_conv2d_fwd_writes_bf16_y(N, C, H, W, C, 3, 1, 1, 0, 0, 0, fl)))


def down_ok(N, C, H, W, Cout, in_bf16):
    """A down-sampling block C -> Cout on (N, C, H, W): every launch in a bf16-reading / -writing matrix-core form."""
    lib, fl = L.load(), L.CONV_FLAGS
    if H % 2 or W % 8 or H < 8:
        return False
    H2, W2 = H // 2, W // 2
    ok = bool(lib.agl_conv2d_fwd_writes_bf16_y(N, C, H, W, C, 3, 1, 1, 0, 1, 0, fl))                       # c1 writes h as bf16
    ok = ok and bool(lib.agl_conv2d_fwd_takes_bf16_x(N, C, H, W, Cout, 4, 2, 1, fl))                        # pooled c2 reads it
    ok = ok and bool(lib.agl_conv2d_bwd_weight_takes_bf16_x(N, C, H, W, Cout, H2, W2, 4, 2, 1, fl))
    ok = ok and bool(lib.agl_conv2d_bwd_data_takes_bf16_mask(N, C, H, W, Cout, H2, W2, 4, 2, 1, fl))        # ... and is its ReLU mask
    ok = ok and bool(lib.agl_conv2d_fwd_packed_bytes(N, C, H2, W2, Cout, 1, 1, 0, 0, fl))                   # shortcut + addend on the patch kernel
    if in_bf16:
        ok = ok and bool(lib.agl_conv2d_fwd_takes_bf16_x(N, C, H, W, C, 3, 1, 1, fl))
        ok = ok and bool(lib.agl_conv2d_bwd_weight_takes_bf16_x(N, C, H, W, C, H, W, 3, 1, 1, fl))
        ok = ok and bool(lib.agl_conv2d_bwd_data_takes_bf16_mask(N, C, H, W, C, H, W, 3, 1, 1, fl))
    return ok


def out_bf16_ok(N, Cin, H2, W2, Cout):
    """The 1x1 shortcut + addend launch of a down block can write the block output as bf16."""
    return bool(L.load().agl_conv2d_fwd_writes_bf16_y(N, Cin, H2, W2, Cout, 1, 1, 0, 0, 0, 0, L.CONV_FLAGS))


_blocked_memo = {}


def blocked_ok(kinds, chans, k0, k1, out16, N, H, W):
    """Every launch of the covered blocks [k0, k1) has its channel-blocked form (C ABI predicates) — all or nothing per call."""
    if not D_BLOCKED or k1 <= k0:
        return False
    key = (tuple(kinds), tuple(chans), k0, k1, tuple(out16), N, H, W, L.CONV_FLAGS)
    hit = _blocked_memo.get(key)
    if hit is None:
        hit = _blocked_memo[key] = _blocked_ok(kinds, chans, k0, k1, out16, N, H, W)
    return hit


def _blocked_ok(kinds, chans, k0, k1, out16, N, H, W):
    lib, fl = L.load(), L.CONV_FLAGS
    XB, YB = L.CONV_X_BF16 | L.CONV_X_BLOCKED, L.CONV_Y_BF16 | L.CONV_Y_BLOCKED
    h, w = H, W
    if kinds[0] == "first_down":
        h, w = H // 2, W // 2
    for k in range(k0, k1):
        cin, cout = chans[k]
        if kinds[k] == "first_flat":
            if not out16[k - k0] or cout % 8 or w % 4:          # (the shortcut launch writes a blocked y only)
                return False
            if not lib.agl_conv2d_fwd_takes_blocked(N, cout, h, w, cout, 3, 1, 1, fl | XB | YB):
                return False
        else:
            if cin % 16 or cout % 8 or w % 8 or h % 2:
                return False
            if not lib.agl_conv2d_fwd_takes_blocked(N, cin, h, w, cin, 3, 1, 1, fl | XB | YB):            # c1
                return False
            if not lib.agl_conv2d_fwd_takes_blocked(N, cin, h, w, cout, 4, 2, 1, fl | XB):                # pooled c2
                return False
            h, w = h // 2, w // 2
            if out16[k - k0] and not lib.agl_conv2d_fwd_takes_blocked(N, cin, h, w, cout, 1, 1, 0, fl | YB):      # shortcut + sum
                return False
    return True


_cover_memo = {}


def cover(kinds: Sequence[str], chans: Sequence[Tuple[int, int]], N: int, H: int, W: int):
    """(first covered block, one past the last, per covered block: output stored as bf16) for a chain whose block k is
    kinds[k] in {"first_flat", "first_down", "down"} with channels chans[k] = (Cin, Cout), on an (N, 3, H, W) input.
    Blocks are taken in order while every launch of the block has a bf16-capable matrix-core form; a block hands bf16 to its
    successor only when the successor can read it; the last covered block hands fp32 to the per-op graph."""
    key = (tuple(kinds), tuple(chans), N, H, W, L.CONV_FLAGS)
    hit = _cover_memo.get(key)
    if hit is not None:
        return hit
    res = (0, 0, [])
    if D_TRUNK and _flags_ok():
        k0, h, w = 0, H, W
        if kinds[0] == "first_down":       # (3 -> C with down-sampling: the image discriminator's stem stays on the per-op graph)
            k0, h, w = 1, H // 2, W // 2
        out16: List[bool] = []
        in16 = False
        k = k0
        while k < len(kinds):
            cin, cout = chans[k]
            if kinds[k] == "first_flat":
                if not flat_first_ok(N, h, w, cout):
                    break
                can16 = True
            else:
                ok = down_ok(N, cin, h, w, cout, in16)
                if not ok and in16 and down_ok(N, cin, h, w, cout, False):
                    ok, in16 = True, False
                    out16[-1] = False          # the predecessor hands over fp32 instead
                if not ok:
                    break
                h, w = h // 2, w // 2
                can16 = out_bf16_ok(N, cin, h, w, cout)
            out16.append(can16)
            in16 = can16
            k += 1
        if out16:
            out16[-1] = False
            res = (k0, k0 + len(out16), out16)
    _cover_memo[key] = res
    return res


class _DTrunk(torch.autograd.Function):
    """blocks[k] = (kind, in_relu); params: six tensors per block (w1, b1, w2, b2, wsc, bsc)."""

    @staticmethod
    def forward(ctx, x, blocks, out16, *params):
        x = F._c(x)
        o = x
        saved, metas = [], []
        srcs = [getattr(p, "_agl_wsrc", None) if p is not None else None for p in params]
        blk = ctx_blk = bool(blocks and blocks[0][0].endswith("+blk"))      # (run() marks the call: channel-blocked tensors inside)
        blocks = tuple((kd.replace("+blk", ""), r) for kd, r in blocks)
        for k, (kind, in_relu) in enumerate(blocks):
            w1, b1, w2, b2, wsc, bsc = [F._c(p) if p is not None else None for p in params[6 * k: 6 * k + 6]]
            s1, s2, ssc = srcs[6 * k], srcs[6 * k + 2], srcs[6 * k + 4]
            ox = None
            if kind == "first_flat":
                h16 = L.conv2d_fwd(o, w1, b1, 1, 1, 0, False, True, wsrc=s1, out_bf16=True, out_blk=blk)
                out = L.conv2d_fwd_shortcut(h16, w2, b2, o, wsc.reshape(wsc.shape[0], -1), bsc, 1, wsrc=s2, out_bf16=out16[k], out_blk=blk and out16[k])
                saved += [o, h16, None, None]
            else:
                C, Cout = L.nchw_shape(o)[1], w2.shape[0]
                ox = (o if L.is_blk(o) else L.to_blocked_dev(o)) if blk else o      # the convolutions' operand (blocked form: one conversion of an fp32 input)
                h16 = L.conv2d_fwd(ox, w1, b1, 1, 1, 0, in_relu, True, wsrc=s1, out_bf16=True, out_blk=blk)
                s = L.avgpool2_fwd(o, in_relu)
                d2 = s2.derived("pool4f") if s2 is not None else None
                if d2 is not None:
                    hp = L.conv2d_fwd(h16, None, b2, 2, 1, wsrc=d2, w_shape=(Cout, C, 4, 4), make_base=lambda w2=w2, s2=s2: _pooled_src(w2, s2))
                else:
                    hp = L.conv2d_fwd(h16, _pooled(w2), b2, 2, 1)
                out = L.conv2d_fwd_addend(s, wsc, bsc, hp, 1, 0, wsrc=ssc, out_bf16=out16[k], out_blk=blk and out16[k])
                saved += [o, h16, s, ox if (blk and ox is not o) else None]
            metas.append((kind, in_relu))
            o = out
        ctx.metas, ctx.srcs, ctx.nparams = metas, srcs, len(params)
        # arena gradient slots of the leaf parameters (the biases: the weights are spectral-norm outputs, their gradients go on to that
        # node): the kernels add into them and the node returns None — no autograd add launch per bias and call
        ctx.slots = [F._slot(p) for p in params]
        ctx.save_for_backward(*([t for t in saved if t is not None] + [p for p in params if p is not None]))
        ctx.layout = ([t is not None for t in saved], [p is not None for p in params])
        return o

    @staticmethod
    def backward(ctx, dout):
        tens = list(ctx.saved_tensors)
        has_s, has_p = ctx.layout
        it = iter(tens)
        saved = [next(it) if h else None for h in has_s]
        params = [next(it) if h else None for h in has_p]
        need = ctx.needs_input_grad
        grads = [None] * ctx.nparams
        d = F._c(dout)
        for k in reversed(range(len(ctx.metas))):
            kind, in_relu = ctx.metas[k]
            o, h16, s, ox = saved[4 * k: 4 * k + 4]
            if ox is None:
                ox = o              # (the convolutions' operand is the block input itself)
            w1, b1, w2, b2, wsc, bsc = params[6 * k: 6 * k + 6]
            s1, s2, ssc = ctx.srcs[6 * k], ctx.srcs[6 * k + 2], ctx.srcs[6 * k + 4]
            nw = [need[3 + 6 * k + j] for j in range(6)]
            sl = ctx.slots[6 * k: 6 * k + 6]
            need_in = k > 0 or need[0]
            H, W = L.nchw_shape(o)[2], L.nchw_shape(o)[3]
            if kind == "first_flat":
                # out = c2(h) + sc(x): the shortcut's gradients from (d, x), the residual branch through h
                dh = L.conv2d_bwd_data(d, w2, (H, W), 1, 1, pos_mask=h16, wsrc=s2)
                grads[6 * k + 2], grads[6 * k + 3] = F._conv_param_grads((sl[2], sl[3]), nw[2], b2 is not None and nw[3], d, h16, w2, 1, 1, 0, False)
                grads[6 * k + 4], grads[6 * k + 5] = F._conv_param_grads((sl[4], sl[5]), nw[4], bsc is not None and nw[5], d, o, wsc, 1, 0, 0, False)
                grads[6 * k + 0], grads[6 * k + 1] = F._conv_param_grads((sl[0], sl[1]), nw[0], b1 is not None and nw[1], dh, o, w1, 1, 1, 0, False)
                if need_in:
                    do = L.conv2d_bwd_data(dh, w1, (H, W), 1, 1, wsrc=s1)
                    L.conv2d_bwd_data(d, wsc, (H, W), 1, 0, out=do, accumulate=True, wsrc=ssc)
                    d = do
                continue
            C, Cout = L.nchw_shape(o)[1], w2.shape[0]
            # shortcut: out = hp + sc(s)
            ds = L.conv2d_bwd_data(d, wsc, (H // 2, W // 2), 1, 0, wsrc=ssc)
            grads[6 * k + 4], grads[6 * k + 5] = F._conv_param_grads((sl[4], sl[5]), nw[4], bsc is not None and nw[5], d, s, wsc, 1, 0, 0, False)
            # residual branch: hp = conv4x4s2(h; pooled w2): input gradient masked by h > 0 (bf16 mask), weight gradient on the 4x4
            # form and back to the 3x3 filter
            dh = L.conv2d_bwd_data(d, None, (H, W), 2, 1, pos_mask=h16, wsrc=s2.derived("pool4") if s2 is not None else None,
                                   w_shape=(Cout, C, 4, 4), make_w=lambda w2=w2: _pooled(w2), make_base=lambda w2=w2, s2=s2: _pooled_src(w2, s2))
            want_b2 = b2 is not None and nw[3]
            if nw[2]:
                db2 = sl[3] if (want_b2 and sl[3] is not None) else (torch.empty(Cout, dtype=torch.float32, device=d.device) if want_b2 else None)
                dw4 = L.conv2d_bwd_weight(d, h16, 4, 2, 1, dbias=db2, dbias_accumulate=want_b2 and sl[3] is not None)
                dw3 = torch.empty((Cout, C, 3, 3), dtype=torch.float32, device=d.device)
                L.call("agl_pool_fuse_weight_bwd", L.ptr(dw4), L.ptr(dw3), Cout * C, L.stream())
                grads[6 * k + 2], grads[6 * k + 3] = dw3, (None if sl[3] is not None else db2)
            elif want_b2:
                if sl[3] is not None:
                    L.channel_sum(d, out=sl[3], accumulate=True)
                else:
                    grads[6 * k + 3] = L.channel_sum(d)
            grads[6 * k + 0], grads[6 * k + 1] = F._conv_param_grads((sl[0], sl[1]), nw[0], b1 is not None and nw[1], dh, ox, w1, 1, 1, 0, in_relu)
            if need_in:
                do = L.avgpool2_bwd(ds, o if in_relu else L.nchw_shape(o), in_relu)
                L.conv2d_bwd_data(dh, w1, (H, W), 1, 1, pos_mask=o if in_relu else None, out=do, accumulate=True, wsrc=s1)
                d = do
        return (d if need[0] else None, None, None) + tuple(grads)


def run(x, blocks, out16, params, blocked=False):
    """blocks: [(kind, in_relu)], out16: per block, params: flat list of 6 tensors per block -> the last block's output.
    blocked: the node keeps its bf16 tensors channel-blocked (blocked_ok said every launch has that form)."""
    if blocked:
        blocks = [(kd + "+blk", r) for kd, r in blocks]
    return _DTrunk.apply(x, tuple(blocks), tuple(out16), *params)
