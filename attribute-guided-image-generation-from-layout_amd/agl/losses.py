"""Loss terms of the reference loop (train64.py:195-245, :284-354) as HIP kernels that return the loss
value (into a device scalar slot) together with coef * dLoss/dInput, so the training step backpropagates
with torch.autograd.backward(outputs, grads) and never builds autograd nodes for the loss arithmetic."""
from __future__ import annotations

import torch

from . import lib as L


def _rows_scratch(rows, device):
    """Own scratch of a row-wise loss call (the row terms must survive until the call's last kernel: not the shared workspace)."""
    return torch.empty(L.load().agl_loss_rows_ws_bytes(rows), dtype=torch.uint8, device=device)


def bce_const(x, target: float, coef: float, slot: torch.Tensor):
    x = x.contiguous()
    dx = torch.empty_like(x)
    L.call("agl_bce_logits_const", L.ptr(x), x.numel(), float(target), float(coef), L.ptr(slot), L.ptr(dx), L.stream())
    return dx


def bce_posw(x, targets, pos_weight, coef: float, slot: torch.Tensor):
    x, targets = x.contiguous(), targets.contiguous()
    rows, A = x.shape
    dx = torch.empty_like(x)
    ws = _rows_scratch(rows, x.device)
    L.call("agl_bce_logits_posw_ws", L.ptr(x), L.ptr(targets), L.ptr(pos_weight), rows, A, float(coef), L.ptr(slot), L.ptr(dx),
           ws.data_ptr(), ws.numel(), L.stream())
    return dx


def cross_entropy(logits, labels, coef: float, slot: torch.Tensor):
    logits = logits.contiguous()
    R, V = logits.shape
    dl = torch.empty_like(logits)
    ws = _rows_scratch(R, logits.device)
    L.call("agl_cross_entropy_ws", L.ptr(logits), L.ptr(labels, torch.int64), R, V, float(coef), L.ptr(slot), L.ptr(dl),
           ws.data_ptr(), ws.numel(), L.stream())
    return dl


def l1_rows(a, b, keep, coef: float, denom: float, slot: torch.Tensor):
    a, b = a.contiguous(), b.contiguous()
    N = a.shape[0] if keep is not None else 1
    ln = a.numel() // N
    da = torch.empty_like(a)
    ws = torch.empty(L.load().agl_l1_rows_ws_bytes(), dtype=torch.uint8, device=a.device)   # own scratch: the partials must
    L.call("agl_l1_rows", L.ptr(a), L.ptr(b), L.ptr(keep), N, ln, float(coef), float(denom), L.ptr(slot), L.ptr(da),   # survive
           ws.data_ptr(), ws.numel(), L.stream())                                                                    # until pass 2
    return da


def kl_sum(mu, logvar, coef: float, slot: torch.Tensor):
    mu, logvar = mu.contiguous(), logvar.contiguous()
    dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
    L.call("agl_kl_sum", L.ptr(mu), L.ptr(logvar), mu.numel(), float(coef), L.ptr(slot), L.ptr(dmu), L.ptr(dlv), L.stream())
    return dmu, dlv


# ---- hinge GAN losses (models/spade/networks/loss.py:65-76).  NOT on the reference's train path: train64.py/train128.py
# use BCE-with-logits (see agl.trainer); BASELINE.json's north_star names `loss_hinge_*`, so they are provided with the
# same kernel contract as the terms above (value into `slot`, returns coef * dLoss/dInput).
_HINGE_MODE = {"dis_real": 0, "dis_fake": 1, "gen": 2}


def hinge(x, mode: str, coef: float, slot: torch.Tensor):
    x = x.contiguous()
    dx = torch.empty_like(x)
    L.call("agl_hinge_loss", L.ptr(x), x.numel(), _HINGE_MODE[mode], float(coef), L.ptr(slot), L.ptr(dx), L.stream())
    return dx


class _HingeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        slot = torch.empty(1, dtype=torch.float32, device=x.device)
        ctx.save_for_backward(hinge(x, mode, 1.0, slot))
        return slot.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None


def loss_hinge_dis(dis_fake, dis_real):
    """GANLoss('hinge') for the discriminator (loss.py:66-72): mean(relu(1 - real)) + mean(relu(1 + fake)) as a
    differentiable scalar.  Off the reference's train path."""
    return _HingeFn.apply(dis_real, "dis_real") + _HingeFn.apply(dis_fake, "dis_fake")


def loss_hinge_gen(dis_fake):
    """GANLoss('hinge') for the generator (loss.py:73-75): -mean(fake).  Off the reference's train path."""
    return _HingeFn.apply(dis_fake, "gen")
