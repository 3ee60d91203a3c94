"""Loss terms of the reference loop (train64.py:195-245, :284-354) as HIP kernels that return the loss
value (into a device scalar slot) together with coef * dLoss/dInput, so the training step backpropagates
with torch.autograd.backward(outputs, grads) and never builds autograd nodes for the loss arithmetic."""
from __future__ import annotations

import torch

from . import lib as L


def bce_const(x, target: float, coef: float, slot: torch.Tensor):
    x = x.contiguous()
    dx = torch.empty_like(x)
    L.call("agl_bce_logits_const", L.ptr(x), x.numel(), float(target), float(coef), L.ptr(slot), L.ptr(dx), L.stream())
    return dx


def bce_posw(x, targets, pos_weight, coef: float, slot: torch.Tensor):
    x, targets = x.contiguous(), targets.contiguous()
    rows, A = x.shape
    dx = torch.empty_like(x)
    L.call("agl_bce_logits_posw", L.ptr(x), L.ptr(targets), L.ptr(pos_weight), rows, A, float(coef), L.ptr(slot), L.ptr(dx), L.stream())
    return dx


def cross_entropy(logits, labels, coef: float, slot: torch.Tensor):
    logits = logits.contiguous()
    R, V = logits.shape
    dl = torch.empty_like(logits)
    L.call("agl_cross_entropy", L.ptr(logits), L.ptr(labels, torch.int64), R, V, float(coef), L.ptr(slot), L.ptr(dl), L.stream())
    return dl


def l1_rows(a, b, keep, coef: float, denom: float, slot: torch.Tensor):
    a, b = a.contiguous(), b.contiguous()
    N = a.shape[0] if keep is not None else 1
    ln = a.numel() // N
    da = torch.empty_like(a)
    L.call("agl_l1_rows", L.ptr(a), L.ptr(b), L.ptr(keep), N, ln, float(coef), float(denom), L.ptr(slot), L.ptr(da), L.stream())
    return da


def kl_sum(mu, logvar, coef: float, slot: torch.Tensor):
    mu, logvar = mu.contiguous(), logvar.contiguous()
    dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
    L.call("agl_kl_sum", L.ptr(mu), L.ptr(logvar), mu.numel(), float(coef), L.ptr(slot), L.ptr(dmu), L.ptr(dlv), L.stream())
    return dmu, dlv
