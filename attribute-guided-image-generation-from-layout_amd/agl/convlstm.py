"""Object-fusing ConvLSTM (reference models/generator_obj_att.py:232-346 LayoutConvLSTM with the
ConvLSTMCell of :63-118), restructured for the GPU without changing its results:

  * conv(cat[x_t, h]) = conv_x(x_t) + conv_h(h): the input half (80 % of layer-0 MACs) is ONE batched
    implicit-GEMM over all O objects per layer instead of O batch-1 launches;
  * the recurrence runs batched over images: images are ordered by decreasing object count, so the
    images still active at step t form a prefix and every per-step operand is a contiguous row slice
    of a time-major buffer (row off[t] + slot);
  * backward is hand-written BPTT on the same buffers: per-step gate/hidden gradients, then ONE
    weight-gradient GEMM per layer for W_h and one for W_x.

The reference runs N images x 3 layers x T steps of batch-1 5x5 convolutions (~1150 launches per
call at N=64); this runs 3 + 3*(T-1) convolutions.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import lib as L


DEFER_SUM = True     # False: every recurrence convolution finishes its own reduction split (tests: identical results, one launch more per step)


class SequencePlan:
    """Host-side bookkeeping derived from the (CPU) obj_to_img vector: run lengths, the image order by
    decreasing length, and the row maps between object-major and time-major layouts."""

    def __init__(self, obj_to_img_cpu: torch.Tensor, device):
        ids = obj_to_img_cpu.detach().cpu().numpy().astype(np.int64).reshape(-1)
        O = ids.shape[0]
        if O == 0:
            raise ValueError("LayoutConvLSTM: empty object list")
        # consecutive runs (the reference splits on every change of image id, generator_obj_att.py:286-304)
        change = np.nonzero(np.diff(ids) != 0)[0] + 1
        first = np.concatenate([[0], change]).astype(np.int64)
        lens = np.diff(np.concatenate([first, [O]])).astype(np.int64)
        N = first.shape[0]
        order = np.argsort(-lens, kind="stable")            # slot -> run index
        slot_of = np.empty(N, np.int64)
        slot_of[order] = np.arange(N)
        T = int(lens.max())
        n_t = np.array([(lens > t).sum() for t in range(T)], np.int64)
        off = np.concatenate([[0], np.cumsum(n_t)]).astype(np.int64)
        tm_to_obj = np.empty(O, np.int64)
        for t in range(T):
            sl = order[: n_t[t]]
            tm_to_obj[off[t]: off[t] + n_t[t]] = first[sl] + t
        last_rows = off[lens - 1] + slot_of                  # per run (original order): row of its final step
        hprev = np.concatenate([off[t - 1] + np.arange(n_t[t]) for t in range(1, T)]) if T > 1 else np.zeros(0, np.int64)
        self.O, self.N, self.T = O, N, T
        self.n_t = [int(v) for v in n_t]
        self.off = [int(v) for v in off]
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.tm_to_obj = up(tm_to_obj)
        self.last_rows = up(last_rows)
        self.hprev_rows = up(hprev)


def _split_w(W, cx):
    """W (4h, cx+h, 5, 5) -> contiguous W[:, :cx], W[:, cx:] (input / recurrent halves)."""
    n, c = W.shape[0], W.shape[1]
    a = torch.empty((n, cx, 5, 5), dtype=torch.float32, device=W.device)
    b = torch.empty((n, c - cx, 5, 5), dtype=torch.float32, device=W.device)
    L.call("agl_concat2_bwd", L.ptr(W.contiguous()), L.ptr(a), L.ptr(b), n, cx, c - cx, 25, 0, L.stream())
    return a, b


def _join_w(dWx, dWh):
    n, ca, cb = dWx.shape[0], dWx.shape[1], dWh.shape[1]
    out = torch.empty((n, ca + cb, 5, 5), dtype=torch.float32, device=dWx.device)
    L.call("agl_concat2_fwd", L.ptr(dWx), L.ptr(dWh), L.ptr(out), n, ca, cb, 25, 0, L.stream())
    return out


class _LayoutConvLSTM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan: SequencePlan, hidden: Sequence[int], *params):
        x = x.contiguous()
        O, _, SH, SW = x.shape
        assert O == plan.O
        S = SH * SW
        dev = x.device
        saved = []
        wsrcs = []          # packed-weight cache handles of the two halves of every layer's weight (agl.lib.WeightSrc)
        X = x
        for li, hid in enumerate(hidden):
            W, b = params[2 * li], params[2 * li + 1]
            cx = X.shape[1]
            Wx, Wh = _split_w(W, cx)
            wsw = getattr(W, "_agl_wsrc", None)
            sx, sh = (wsw.derived("x"), wsw.derived("h")) if wsw is not None else (None, None)
            wsrcs.append((sx, sh))
            ccx = L.conv2d_fwd(X, Wx, b, 1, 2, wsrc=sx)                           # (O, 4h, 8, 8)
            H = torch.empty((O, hid, SH, SW), dtype=torch.float32, device=dev)    # time-major
            Cs = torch.empty_like(H)
            gates = torch.empty((O, 4 * hid, SH, SW), dtype=torch.float32, device=dev)
            for t in range(plan.T):
                n, o = plan.n_t[t], plan.off[t]
                if li == 0:
                    src, rows = ccx, plan.tm_to_obj[o:o + n]
                else:
                    src, rows = ccx[o:o + n], None
                if t == 0:
                    cch = cprev = None
                else:
                    po = plan.off[t - 1]
                    # (a reduction split's partial outputs are added by the gate kernel: no epilogue launch between the two)
                    cch = L.conv2d_fwd(H[po:po + n], Wh, None, 1, 2, wsrc=sh, defer=DEFER_SUM)
                    cprev = Cs[po:po + n]
                L.lstm_gates_fwd(src, rows, cch, cprev, H[o:o + n], Cs[o:o + n], gates[o:o + n], n, hid, S)
            saved += [X, Wx, Wh, H, Cs, gates]
            X = H
        out = L.gather_rows(X, plan.last_rows)
        ctx.plan, ctx.hidden, ctx.wsrcs = plan, tuple(hidden), wsrcs
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        plan, hidden = ctx.plan, ctx.hidden
        saved = ctx.saved_tensors
        dout = dout.contiguous()
        nl = len(hidden)
        SH, SW = dout.shape[2], dout.shape[3]
        S = SH * SW
        dev = dout.device
        dH_ext = torch.zeros((plan.O, hidden[-1], SH, SW), dtype=torch.float32, device=dev)
        L.scatter_rows(dout, plan.last_rows, dH_ext)
        grads: List = [None] * (2 * nl)
        dX = None
        for li in reversed(range(nl)):
            hid = hidden[li]
            X, Wx, Wh, H, Cs, gates = saved[6 * li: 6 * li + 6]
            sx, sh = ctx.wsrcs[li]
            dCC = torch.empty_like(gates)
            dc_carry = dh_rec = None
            n_next = 0
            for t in reversed(range(plan.T)):
                n, o = plan.n_t[t], plan.off[t]
                dc_prev = torch.empty((n, hid, SH, SW), dtype=torch.float32, device=dev)
                cprev = Cs[plan.off[t - 1]: plan.off[t - 1] + n] if t > 0 else None
                L.lstm_gates_bwd(dH_ext[o:o + n], dh_rec, n_next, dc_carry, n_next, gates[o:o + n], cprev, Cs[o:o + n],
                                 dCC[o:o + n], dc_prev, n, hid, S)
                dc_carry = dc_prev
                # (consumed by the NEXT iteration's gate kernel, the next launch on this stream: its slabs may stay unreduced)
                dh_rec = L.conv2d_bwd_data(dCC[o:o + n], Wh, (SH, SW), 1, 2, wsrc=sh, defer=DEFER_SUM) if t > 0 else None
                n_next = n
            n0 = plan.n_t[0]
            if plan.T > 1:
                Hprev = L.gather_rows(H, plan.hprev_rows)
                dWh = L.conv2d_bwd_weight(dCC[n0:], Hprev, 5, 1, 2)
            else:
                dWh = torch.zeros_like(Wh)
            if li == 0:
                dCCx = torch.empty_like(dCC)
                L.scatter_rows(dCC, plan.tm_to_obj, dCCx)         # back to object-major
            else:
                dCCx = dCC
            dWx = L.conv2d_bwd_weight(dCCx, X, 5, 1, 2)
            grads[2 * li] = _join_w(dWx, dWh)
            grads[2 * li + 1] = L.channel_sum(dCCx)
            if li > 0 or ctx.needs_input_grad[0]:
                dX = L.conv2d_bwd_data(dCCx, Wx, (SH, SW), 1, 2, wsrc=sx)
            dH_ext = dX
        return (dX if ctx.needs_input_grad[0] else None, None, None) + tuple(grads)


def layout_conv_lstm(x, plan: SequencePlan, hidden: Sequence[int], weights: Sequence[torch.Tensor],
                     biases: Sequence[torch.Tensor]):
    params = []
    for w, b in zip(weights, biases):
        params += [w, b]
    return _LayoutConvLSTM.apply(x, plan, tuple(hidden), *params)
