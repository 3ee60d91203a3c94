"""ORACLE — TEST INFRASTRUCTURE ONLY.

Closed-form, RNG-free parameter fill so that 30 M-parameter networks never have to be
shipped as fixtures: both the reference (in the build container) and the product / the
oracle (anywhere) rebuild identical weights from this rule, keyed only by the
state_dict key order and tensor shapes (SURVEY.md §8c "Param fill rule").
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch


def _wave(n: int, k: int) -> np.ndarray:
    """n pseudo-random values in [-1, 1) from a splitmix64 integer hash of (tensor index k, element index):
    closed-form and platform independent, but statistically like the reference's default uniform init, so
    the filled networks are as well conditioned in fp32 as freshly initialised ones (a smooth sin() fill
    made the generator's fp32 result itself uncertain to ~1e-2)."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64((k + 1) * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def fill_state(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Return a new state dict (same keys/shapes/dtypes) with deterministic values."""
    out = {}
    for k, (name, t) in enumerate(state.items()):
        n = t.numel()
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros_like(t)
            continue
        wv = _wave(n, k)
        if name.endswith("running_mean"):
            v = 0.05 * wv
        elif name.endswith("running_var"):
            v = 1.0 + 0.2 * wv
        elif name.endswith(("weight_u", "weight_v")):
            v = wv + 0.3
            v = v / max(np.linalg.norm(v), 1e-12)
        elif name.endswith("embed.weight"):                      # CondBN: [gamma | beta]
            C = shape[1] // 2
            v = wv.reshape(shape)
            v = np.concatenate([1.0 + 0.1 * v[:, :C], 0.1 * v[:, C:]], axis=1)
        elif name.endswith("embedding.weight"):
            v = wv
        elif name.endswith("bias"):
            v = 0.05 * wv
        elif len(shape) == 1:                                     # BN affine weight
            v = 1.0 + 0.1 * wv
        else:                                                     # conv / linear weight
            fan_in = n // shape[0]
            v = math.sqrt(1.0 / fan_in) * wv
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float64).reshape(shape)).to(t.dtype)
    return out
