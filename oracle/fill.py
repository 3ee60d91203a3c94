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
    i = np.arange(n, dtype=np.float64)
    return np.sin(0.37 * i + 0.11 * k + 0.5 * np.sin(0.013 * i + k))


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
            v = math.sqrt(3.0 / fan_in) * wv
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float64).reshape(shape)).to(t.dtype)
    return out
