"""Spectral-normalised ResNet discriminators (image / object / attribute) on the HIP kernels.

Names, constructor arguments and state_dict layout follow reference models/discriminator.py
(:29-60 OptimizedBlock, :63-99 ResidualBlock, :102-181 AttributeDiscriminator[128], :184-230
ImageDiscriminator, :233-278 ObjectDiscriminator, :15-22 add_sn).  Differences are legal algebraic
fusions only: bias+ReLU in the conv epilogue, the block-leading in-place ReLU folded into the conv
gathers (so the shortcut also reads relu(x), as the reference's aliasing makes it), the 1x1 shortcut
conv applied after the 2x2 average pool and accumulated straight into the residual branch, and one
batched spectral-norm kernel sequence per forward call.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import dtrunk as T
from . import functional as F
from . import lib as L
from . import nn as A


def add_sn(m):
    """Reference add_sn (discriminator.py:15-22): children first, then the module itself."""
    for name, c in m.named_children():
        m.add_module(name, add_sn(c))
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        return A.apply_spectral_norm(m)
    if isinstance(m, (nn.ConvTranspose2d, nn.Embedding)):
        raise NotImplementedError("spectral norm on ConvTranspose2d/Embedding is not on the reference path")
    return m


def _w(W, mod):
    return None if W is None else W.get(id(mod))


def _fork(x, c1, w1, in_relu):
    """First convolution of a down-sampling block and the pooled shortcut input, from the block input x."""
    w = c1.weight if w1 is None else w1
    if FORK:
        return F.conv2d_and_avg_pool2(x, w, c1.bias, c1.padding[0], in_relu=in_relu, relu=True, relu_grad_by_consumer=True)
    return c1(x, in_relu=in_relu, relu=True, relu_grad_by_consumer=True, weight=w1), F.avg_pool2(x, in_relu=in_relu)


FORK = True      # False: two graph nodes, autograd adds their input gradients (closed A/B, round 3; tests flip it in process)


class OptimizedBlock(nn.Module):
    def __init__(self, dim_in, dim_out, downsample=False):
        super().__init__()
        self.downsample = downsample
        self.resi = nn.Sequential(A.Conv2d(dim_in, dim_out, kernel_size=3, stride=1, padding=1, bias=True),
                                  nn.ReLU(inplace=True),
                                  A.Conv2d(dim_out, dim_out, kernel_size=3, stride=1, padding=1, bias=True))
        self.learnable_sc = (dim_in != dim_out) or downsample
        if self.learnable_sc:
            self.sc = A.Conv2d(dim_in, dim_out, kernel_size=1, padding=0, bias=True)

    def forward(self, x, W=None):
        # h feeds exactly one convolution, which masks its input gradient by h > 0 (no separate ReLU-backward pass)
        c1, c2 = self.resi[0], self.resi[2]
        w1, w2 = _w(W, c1), _w(W, c2)
        if self.downsample:            # conv3x3 + avg-pool == one 4x4 stride-2 conv (2.25x fewer MACs)
            # (x feeds the first convolution and the pooled shortcut: one graph node, its input gradient is accumulated in place)
            h, s = _fork(x, c1, w1, False)
            h = F.conv3x3_avgpool2(h, c2.weight if w2 is None else w2, c2.bias, x_relu=True)
        else:
            # bf16 arithmetic: h = relu(c1(x)) feeds c2 only — the two convolutions as one graph node, h stored as bf16 inside it
            # (F.conv_relu_conv3x3) where the kernels that run can write / read that form
            N, C0, H, Wd = x.shape
            wa, wb = (w1 if w1 is not None else c1.weight), (w2 if w2 is not None else c2.weight)
            if F.H_BF16 and L.first_conv_output_as_bf16(N, C0, H, Wd, c1.out_channels, c2.out_channels, c1.kernel_size[0],
                                                        need_bww=wb.requires_grad, need_bwd_data=True):
                h = F.conv_relu_conv3x3(x, wa, c1.bias, wb, c2.bias, c1.padding[0], h_bf16=True)
            else:
                h = c1(x, relu=True, relu_grad_by_consumer=True, weight=w1)
                h = c2(h, x_relu=True, weight=w2)
            s = x
        return self.sc(s, addend=h, weight=_w(W, self.sc))


class ResidualBlock(nn.Module):
    def __init__(self, dim_in, dim_out, downsample=False):
        super().__init__()
        self.downsample = downsample
        self.resi = nn.Sequential(nn.ReLU(inplace=True),
                                  A.Conv2d(dim_in, dim_in, kernel_size=3, stride=1, padding=1, bias=True),
                                  nn.ReLU(inplace=True),
                                  A.Conv2d(dim_in, dim_out, kernel_size=3, stride=1, padding=1, bias=True))
        self.learnable_sc = (dim_in != dim_out) or downsample
        if self.learnable_sc:
            self.sc = A.Conv2d(dim_in, dim_out, kernel_size=1, padding=0, bias=True)

    def forward(self, x, W=None):
        if not self.learnable_sc:
            raise NotImplementedError("identity-shortcut blocks are not on the reference path")
        c1, c2 = self.resi[1], self.resi[3]
        w1, w2 = _w(W, c1), _w(W, c2)
        if self.downsample:
            h, s = _fork(x, c1, w1, True)
            h = F.conv3x3_avgpool2(h, c2.weight if w2 is None else w2, c2.bias, x_relu=True)
            return self.sc(s, addend=h, weight=_w(W, self.sc))
        h = c1(x, in_relu=True, relu=True, relu_grad_by_consumer=True, weight=w1)
        h = c2(h, x_relu=True, weight=w2)
        return self.sc(x, in_relu=True, addend=h, weight=_w(W, self.sc))


class _Discriminator(nn.Module):
    def _sn_modules(self):
        return [m for m in self.modules() if getattr(m, "has_sn", False)]

    def _weights(self):
        mods = self._sn_modules()
        if not mods:
            return None
        ws = F.spectral_norm_weights([m.weight_orig for m in mods], [m.weight_u for m in mods],
                                     [m.weight_v for m in mods], self.training)
        return {id(m): w for m, w in zip(mods, ws)}

    def _trunk(self, x, W):
        A._need_device(x)
        blocks = list(self.main)
        k0 = k1 = 0
        if W is not None and x.dim() == 4 and x.shape[1] == 3:
            # bf16 arithmetic: a prefix of the block chain as ONE graph node with bf16-stored activations inside (agl.dtrunk)
            kinds = [("first_down" if b.downsample else "first_flat") if isinstance(b, OptimizedBlock) else "down" for b in blocks]
            if all(b.downsample for b in blocks[1:]) and all(getattr(b, "learnable_sc", False) for b in blocks):
                chans = [(b.resi[0].in_channels, b.resi[2].out_channels) if isinstance(b, OptimizedBlock)
                         else (b.resi[1].in_channels, b.resi[3].out_channels) for b in blocks]
                k0, k1, out16 = T.cover(kinds, chans, x.shape[0], x.shape[2], x.shape[3])
        h = x
        for blk in blocks[:k0]:
            h = blk(h, W)
        if k1 > k0:
            spec, params = [], []
            for b, kind in zip(blocks[k0:k1], kinds[k0:k1]):
                c1, c2 = (b.resi[0], b.resi[2]) if isinstance(b, OptimizedBlock) else (b.resi[1], b.resi[3])
                spec.append((kind, not isinstance(b, OptimizedBlock)))
                for m in (c1, c2, b.sc):
                    w = _w(W, m)
                    params += [w if w is not None else m.weight, m.bias]
            blocked = T.blocked_ok(kinds, chans, k0, k1, out16, x.shape[0], x.shape[2], x.shape[3])
            h = T.run(h, spec, out16, params, blocked=blocked)
        for blk in blocks[k1:]:
            h = blk(h, W)
        return F.sum_hw(h, in_relu=True)


class ImageDiscriminator(_Discriminator):
    def __init__(self, conv_dim=64):
        super().__init__()
        self.ch = conv_dim
        c = conv_dim
        self.relu = nn.ReLU(inplace=True)
        self.main = nn.Sequential(OptimizedBlock(3, c, downsample=True), ResidualBlock(c, c * 2, downsample=True),
                                  ResidualBlock(c * 2, c * 4, downsample=True), ResidualBlock(c * 4, c * 8, downsample=True),
                                  ResidualBlock(c * 8, c * 16, downsample=True))
        self.classifier = A.Linear(c * 16, 1, bias=False)

    def forward(self, x, W=None):
        """W: the normalised weights of THIS call when the caller computed them ahead (`net._weights()`, in call order — the
        spectral-norm state advances there); the convolutions may then run on another stream than the network's earlier calls."""
        W = self._weights() if W is None else W
        f = self._trunk(x, W)
        return self.classifier(f, weight=_w(W, self.classifier)).view(-1)


class ObjectDiscriminator(_Discriminator):
    def __init__(self, conv_dim=64, n_class=0, downsample_first=False, n_attribute=128):
        super().__init__()
        c = conv_dim
        self.relu = nn.ReLU(inplace=True)
        self.main = nn.Sequential(OptimizedBlock(3, c, downsample=downsample_first), ResidualBlock(c, c * 2, downsample=True),
                                  ResidualBlock(c * 2, c * 4, downsample=True), ResidualBlock(c * 4, c * 8, downsample=True),
                                  ResidualBlock(c * 8, c * 16, downsample=True))
        self.classifier_src = A.Linear(c * 16, 1)
        self.classifier_cls = A.Linear(c * 16, n_class)

    def forward(self, x, y=None, W=None):
        W = self._weights() if W is None else W
        f = self._trunk(x, W)
        src = self.classifier_src(f, weight=_w(W, self.classifier_src)).view(-1)
        return src, self.classifier_cls(f, weight=_w(W, self.classifier_cls))


class AttributeDiscriminator(_Discriminator):
    n_extra = 0

    def __init__(self, conv_dim=64, downsample_first=False, n_attribute=128):
        super().__init__()
        c = conv_dim
        self.relu = nn.ReLU(inplace=True)
        blocks = [OptimizedBlock(3, c, downsample=downsample_first), ResidualBlock(c, c * 2, downsample=True),
                  ResidualBlock(c * 2, c * 4, downsample=True), ResidualBlock(c * 4, c * 8, downsample=True),
                  ResidualBlock(c * 8, c * 16, downsample=True)]
        blocks += [ResidualBlock(c * 16, c * 16, downsample=True) for _ in range(self.n_extra)]
        self.main = nn.Sequential(*blocks)
        self.classifier_att = A.Linear(c * 16, n_attribute)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        W = self._weights()
        return self.classifier_att(self._trunk(x, W), weight=_w(W, self.classifier_att))


class AttributeDiscriminator128(AttributeDiscriminator):
    n_extra = 1
