"""Host logic either side of the train step, with the heavy parts on device.

  * `estimate_attributes`  train64.py:156-166 — one kernel instead of a Python loop over all objects.
  * `swap_attributes`      train64.py:170-188 — the draws still come from python's `random` (same calls, same
    order, so a seeded run picks the same attributes as the reference); what changes is the data movement: ONE
    device->host copy of the rows that can change and ONE scatter of the replacement rows, instead of a
    device sync per object.
  * `imagenet_deprocess_batch`  data/utils.py:47-66 — de-normalise / rescale / to-bytes in one kernel, bytes equal.
"""
from __future__ import annotations

import math
import random as _random
from typing import Optional

import numpy as np
import torch

from . import lib as L

IMAGENET_MEAN = [0.485, 0.456, 0.406]          # data/utils.py:21-22
IMAGENET_STD = [0.229, 0.224, 0.225]


def estimate_attributes(att_logits: torch.Tensor, attribute: torch.Tensor) -> torch.Tensor:
    """attribute_est: rows without any annotated attribute get the arg-max attribute of D_att's logits."""
    return L.attr_estimate(att_logits, attribute)


def swap_rows(obj_to_img_cpu: torch.Tensor, n_images: int):
    """Object rows whose attributes are redrawn: in each of the first floor(N/3) images, the first
    floor(P/2) objects (train64.py:171-178)."""
    ids = obj_to_img_cpu.detach().cpu().numpy().reshape(-1)
    rows = []
    for img in range(math.floor(n_images / 3)):
        idx = np.nonzero(ids == img)[0]
        rows.extend(int(v) for v in idx[: math.floor(len(idx) / 2)])
    return rows


def swap_attributes(attribute: torch.Tensor, attribute_est: torch.Tensor, objs: torch.Tensor, obj_to_img_cpu: torch.Tensor,
                    matrix: torch.Tensor, n_images: int, rng=_random) -> torch.Tensor:
    """In place on the device tensors `attribute` and `attribute_est` ((O, A) fp32); returns the row indices changed.
    `matrix` is the (V, A) object-vs-attribute co-occurrence table on the CPU (train64.py:83)."""
    A = attribute.shape[1]
    rows = swap_rows(obj_to_img_cpu, n_images)
    if not rows:
        return torch.zeros(0, dtype=torch.int64)
    rows_t = torch.tensor(rows, dtype=torch.int64)
    rows_dev = rows_t.to(attribute.device)
    old = attribute.index_select(0, rows_dev).cpu()               # attribute_GT rows (:153), one copy
    cls = objs.index_select(0, rows_dev).cpu()
    new_rows = torch.zeros((len(rows), A), dtype=torch.float32)
    for i in range(len(rows)):
        old_attributes = torch.nonzero(old[i]).view(-1)
        weights = matrix[int(cls[i])].scatter(0, old_attributes, 0)
        picked = rng.choices(range(A), weights, k=rng.randrange(1, 3))   # same calls as :181-182
        new_rows[i, picked] = 1.0
    new_dev = new_rows.to(attribute.device)
    L.scatter_rows(new_dev, rows_dev, attribute)
    L.scatter_rows(new_dev, rows_dev, attribute_est)
    return rows_t


def imagenet_deprocess_batch(imgs: torch.Tensor, rescale: bool = True) -> torch.Tensor:
    """(N,3,H,W) fp32 on device -> CPU ByteTensor in [0,255], like the reference (which works on a CPU clone)."""
    inv_std = [float(np.float32(1.0 / s)) for s in IMAGENET_STD]
    mean = [float(np.float32(m)) for m in IMAGENET_MEAN]
    return L.deprocess_u8(imgs.detach().float(), inv_std, mean, rescale).cpu()
