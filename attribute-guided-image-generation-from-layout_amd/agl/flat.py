"""Flat fp32 arenas for parameters, gradients and Adam moments.

Each network's parameters become views into ONE contiguous buffer (same for the gradients), so the
optimiser is a single fused kernel launch and the data-parallel gradient exchange is a single RCCL
all-reduce on the arena itself — no per-tensor launches, no bucket copies (HBM is plentiful on
MI355X; launches and small collectives are what cost).
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.nn as nn

from . import lib as L


FUSED_REPACK = True     # False: one pack launch per weight and form at first use instead of the arena-wide re-pack after Adam

class FlatParams:
    def __init__(self, modules: Iterable[nn.Module]):
        self.modules = list(modules)
        self.params: List[nn.Parameter] = [p for m in self.modules for p in m.parameters()]
        assert self.params, "no parameters"
        # Arena order.  The optimiser and the gradient exchange are elementwise, so the arena need not follow .parameters()
        # order: sub-modules may ask (agl_param_pairs) for two parameters to lie back to back, so that their concatenation along
        # dim 0 is a zero-copy VIEW of the arena with a gradient slot of its own (SPADE's gamma|beta convolution: no concat
        # kernels in forward / backward and no autograd add per half).
        pairs = [(mod, name, a, b) for m in self.modules for mod in m.modules() if hasattr(mod, "agl_param_pairs")
                 for (name, a, b) in mod.agl_param_pairs()]
        second_of = {id(b): a for (_, _, a, b) in pairs}
        first_of = {id(a): b for (_, _, a, b) in pairs}
        order: List[nn.Parameter] = []
        for p in self.params:
            if id(p) in second_of:
                continue                    # placed right behind its partner
            order.append(p)
            if id(p) in first_of:
                order.append(first_of[id(p)])
        assert len(order) == len(self.params) and len({id(p) for p in order}) == len(order)
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n = n
        self.p = torch.empty(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.pack_plan = L.PackPlan() if FUSED_REPACK else None
        self.step_count = 0
        self.epoch = 0      # bumped whenever the arena is written through a raw pointer or an alias (Adam kernel, broadcast): the
                            # packed-weight caches of the parameters (agl.lib.WeightSrc) are keyed by (epoch, tensor version)
        o = 0
        offset = {}
        with torch.no_grad():
            for p in order:
                k = p.numel()
                offset[id(p)] = o
                self.p[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.p[o:o + k].view(p.shape)
                p.grad = self.g[o:o + k].view(p.shape)
                p._agl_slot = True          # backward kernels accumulate straight into this slot (functional._slot)
                p._agl_flat, p._agl_off = self, o     # (for the private gradient arenas of concurrent branches, functional.GRAD_ARENA)
                if p.dim() == 4:            # convolution weights: handle of their packed forms
                    p._agl_wsrc = L.WeightSrc(p, (lambda p=p: (self.epoch, p._version)))
                o += k
        for mod, name, a, b in pairs:       # joined leaves: data and gradient alias the two neighbours' arena ranges
            oa, k = offset[id(a)], a.numel() + b.numel()
            assert offset[id(b)] == oa + a.numel() and a.shape[1:] == b.shape[1:]
            shape = (a.shape[0] + b.shape[0],) + tuple(a.shape[1:])
            j = self.p[oa:oa + k].view(shape).detach().requires_grad_(True)
            j.grad = self.g[oa:oa + k].view(shape)
            j._agl_slot = True
            j._agl_flat, j._agl_off = self, oa
            if j.dim() == 4:
                j._agl_wsrc = L.WeightSrc(j, (lambda a=a, b=b: (self.epoch, a._version, b._version)))
            mod.__dict__.setdefault("_agl_joined", {})[name] = j

    def zero_grad(self):
        self.g.zero_()
        for p in self.params:           # re-attach in case autograd replaced a .grad tensor
            if p.grad is None or p.grad.data_ptr() < self.g.data_ptr() or p.grad.data_ptr() >= self.g.data_ptr() + 4 * self.n:
                raise RuntimeError("a parameter's .grad left the flat arena")

    def set_requires_grad(self, flag: bool):
        for p in self.params:
            p.requires_grad_(flag)

    def branch_arenas(self, k: int):
        """k zero-filled buffers laid out like .g (allocated once): private gradient arenas of concurrent branches."""
        if not hasattr(self, "_branch_g") or len(self._branch_g) < k:
            self._branch_g = [torch.zeros_like(self.g) for _ in range(k)]
        return self._branch_g[:k]

    def fold_branch_arenas(self):
        """g += every branch arena; the arenas are zeroed for their next use (all on the current stream)."""
        for b in getattr(self, "_branch_g", []):
            L.axpby(self.g, b, 1.0, 1.0, out=self.g)
            b.zero_()

    def touch(self):
        """Call after writing the arena through anything but the parameters themselves (e.g. a broadcast into .p)."""
        self.epoch += 1

    def adam_step(self, lr, beta1, beta2, eps, grad_scale=1.0):
        """One Adam launch over the arena on the current stream, then the arena's packed weights re-made in place (agl.lib.PackPlan).
        PRECONDITION (the caller's job, as for any in-place optimiser): every launch that reads this arena's weights or their packed
        forms — on any stream — is ordered before the current stream.  Trainer joins all chains and weight-gradient streams first."""
        self.step_count += 1
        self.epoch += 1
        L.adam_step(self.p, self.g, self.m, self.v, lr, beta1, beta2, eps, self.step_count, grad_scale)
        if self.pack_plan is not None:      # every packed form of the arena's weights, in one launch on this stream (agl.lib.PackPlan)
            self.pack_plan.repack()
