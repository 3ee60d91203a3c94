"""Parameter containers with the reference's names / init / state_dict layout whose forward passes run
on the HIP kernels (agl.functional).  Sub-classing the torch containers keeps construction-time RNG
consumption, `state_dict()` keys and `.parameters()` order identical to the reference modules, so a
reference checkpoint loads unchanged and the same seed yields the same initial weights.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F


def _need_device(t):
    if not t.is_cuda:
        raise RuntimeError("attribute-guided-image-generation-from-layout_amd runs on MI355X HIP kernels only: "
                           "move the module and its inputs to a 'cuda' (HIP) device; there is no CPU fallback.")


class Conv2d(nn.Conv2d):
    """nn.Conv2d container; square kernels 1/3/4/5/7, stride 1/2 (all the path uses)."""

    def forward(self, x, *, in_relu=False, relu=False, up=0, addend=None, weight=None, relu_grad_by_consumer=False, x_relu=False):
        _need_device(x)
        w = self.weight if weight is None else weight
        return F.conv2d(x, w, self.bias, self.stride[0], self.padding[0], up, in_relu, relu, addend, relu_grad_by_consumer, x_relu)


class Linear(nn.Linear):
    def forward(self, x, *, relu=False, weight=None):
        _need_device(x)
        return F.linear(x, self.weight if weight is None else weight, self.bias, relu)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        _need_device(x)
        assert self.kernel_size == (4, 4) and self.stride == (2, 2) and self.padding == (1, 1) and self.bias is None
        return F.conv_transpose2d_k4s2p1(x, self.weight)


class BatchNorm2d(nn.BatchNorm2d):
    def forward(self, x, *, relu=False, residual=None):
        _need_device(x)
        return F.batch_norm(x, self.running_mean, self.running_var, self.num_batches_tracked,
                            self.weight if self.affine else None, self.bias if self.affine else None,
                            relu, residual, self.training)


class BatchNorm1d(nn.BatchNorm1d):
    def forward(self, x, *, relu=False):
        _need_device(x)
        y = F.batch_norm(x.reshape(x.shape[0], x.shape[1], 1, 1), self.running_mean, self.running_var,
                         self.num_batches_tracked, self.weight, self.bias, relu, None, self.training)
        return y.reshape(x.shape)


class Embedding(nn.Embedding):
    def forward(self, idx):
        return F.embedding(self.weight, idx)


# ---- spectral norm state on Conv2d / Linear containers (reference models/discriminator.py:15-22) ----
def apply_spectral_norm(m: nn.Module) -> nn.Module:
    """Give a Conv2d/Linear container torch.nn.utils.spectral_norm's state layout: parameter
    `weight_orig` (registered after `bias`), buffers `weight_u`, `weight_v`, initialised with the same
    RNG draws as torch (normalised N(0,1) vectors).  The normalised weight itself is produced per
    forward call by the owning discriminator in one batched kernel sequence."""
    w = m.weight
    h = w.shape[0]
    wd = w.numel() // h
    with torch.no_grad():
        u = TF.normalize(w.new_empty(h).normal_(0, 1), dim=0, eps=F.SN_EPS)
        v = TF.normalize(w.new_empty(wd).normal_(0, 1), dim=0, eps=F.SN_EPS)
    del m._parameters["weight"]
    m.register_parameter("weight_orig", w)
    m.register_buffer("weight_u", u)
    m.register_buffer("weight_v", v)
    m.has_sn = True
    return m
