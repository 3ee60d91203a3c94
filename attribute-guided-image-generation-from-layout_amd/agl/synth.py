"""Synthetic Visual-Genome-shaped batches (host logic, numpy only).

Shapes and value rules follow the reference's dataset + collate code
(data/vg_custom_mask.py:115-173 per-object boxes/masks/shift rule, :176-221 collate):
flat per-object tensors plus a sorted `obj_to_img`.  The mask is the box rasterised with
python `round` (:136); the shifted box moves by 0.8 x the larger horizontal border
distance when the box is narrower than half the image (:140-158).

The random attribute swap of train64.py:170-188 is emulated here (first floor(N/3) images,
first floor(P/2) objects get 1..2 new attributes), so a batch carries `attribute`
(post-swap), `attribute_gt` (pre-swap) and `attribute_est`.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

NUM_OBJECT_CLASSES = 179      # data/vocab.json object_idx_to_name (index 0 = __image__)
NUM_ATTRIBUTES = 106          # train64.py:89


def _raster(mask, y0, y1, x0, x1, R):
    mask[0, round(y0 * R):round(y1 * R), round(x0 * R):round(x1 * R)] = 1.0


def balanced_object_counts(images_per_rank: int, world: int, seed: int = 1234):
    """Object counts per image for `world` data-parallel ranks of `images_per_rank` images each, drawn like make_batch draws them
    (P ~ U{3..9}) for the GLOBAL batch and then dealt to the ranks so that every rank gets the same number of images and (nearly) the
    same number of objects: the object-level work (crop / layout encoders, object and attribute discriminators, ~60 % of the FLOPs)
    scales with the object count and a data-parallel step takes as long as its slowest rank.  Greedy: images by decreasing object
    count, each to the rank with the fewest objects among those that still have room (longest-processing-time rule).  Deterministic in
    (images_per_rank, world, seed); every rank computes the same table and takes its row."""
    rng = np.random.default_rng(seed)
    P = rng.integers(3, 10, size=images_per_rank * world)
    order = np.argsort(-P, kind="stable")
    load = np.zeros(world, np.int64)
    rows = [[] for _ in range(world)]
    for i in order:
        free = [r for r in range(world) if len(rows[r]) < images_per_rank]
        r = min(free, key=lambda q: (load[q], q))
        rows[r].append(int(P[i]))
        load[r] += int(P[i])
    return [np.asarray(sorted(row, reverse=True), dtype=np.int64) for row in rows]


def make_batch(n_images: int, image_size: int, *, seed: int = 1234, z_dim: int = 64,
               objs_per_image=None, n_attr: int = NUM_ATTRIBUTES,
               n_classes: int = NUM_OBJECT_CLASSES) -> Dict[str, np.ndarray]:
    """Return a dict of numpy arrays with the reference's 8-tuple batch schema
    (data/vg_custom_mask.py:219) plus z / attribute_gt / attribute_est."""
    rng = np.random.default_rng(seed)
    R = image_size
    if objs_per_image is None:
        P = rng.integers(3, 10, size=n_images)           # 3..9 objects per image
    elif np.ndim(objs_per_image) == 0:
        P = np.full(n_images, int(objs_per_image), dtype=np.int64)
    else:
        P = np.asarray(objs_per_image, dtype=np.int64)
        assert P.shape == (n_images,)
    O = int(P.sum())
    obj_to_img = np.repeat(np.arange(n_images, dtype=np.int64), P)
    objs = rng.integers(1, n_classes, size=O).astype(np.int64)
    x0 = rng.uniform(0.0, 0.6, size=O)
    y0 = rng.uniform(0.0, 0.6, size=O)
    w = rng.uniform(0.1, 0.4, size=O)
    h = rng.uniform(0.1, 0.4, size=O)
    x1 = np.minimum(x0 + w, 1.0)
    y1 = np.minimum(y0 + h, 1.0)
    boxes = np.stack([x0, y0, x1, y1], axis=1).astype(np.float32)
    boxes_shift = boxes.copy()
    masks = np.zeros((O, 1, R, R), np.float32)
    masks_shift = np.zeros((O, 1, R, R), np.float32)
    for i in range(O):
        bx0, by0, bx1, by1 = (float(v) for v in boxes[i])
        _raster(masks[i], by0, by1, bx0, bx1, R)
        sx0, sx1 = bx0, bx1
        if bx1 - bx0 < 0.5:
            left, right = bx0, 1 - bx1
            if left > right:
                sx0, sx1 = bx0 - 0.8 * left, bx1 - 0.8 * left
            elif right > left:
                sx0, sx1 = bx0 + 0.8 * right, bx1 + 0.8 * right
        _raster(masks_shift[i], by0, by1, sx0, sx1, R)
        boxes_shift[i] = (sx0, by0, sx1, by1)

    attribute_gt = np.zeros((O, n_attr), np.float32)
    has = rng.random(O) < 0.5
    for i in np.nonzero(has)[0]:
        k = int(rng.integers(1, 4))
        attribute_gt[i, rng.choice(n_attr, size=k, replace=False)] = 1.0
    attribute_est = attribute_gt.copy()
    for i in np.nonzero(~has)[0]:                          # stand-in for the argmax estimate
        attribute_est[i, int(rng.integers(0, n_attr))] = 1.0
    attribute = attribute_gt.copy()
    n_swap = math.floor(n_images / 3)
    first = np.concatenate([[0], np.cumsum(P)[:-1]])
    for img in range(n_swap):
        for j in range(math.floor(int(P[img]) / 2)):
            o = int(first[img]) + j
            new = rng.choice(n_attr, size=int(rng.integers(1, 3)), replace=False)
            attribute[o] = 0
            attribute[o, new] = 1
            attribute_est[o] = attribute[o]
    imgs = rng.standard_normal((n_images, 3, R, R)).astype(np.float32)
    z = rng.standard_normal((O, z_dim)).astype(np.float32)
    return dict(imgs=imgs, objs=objs, boxes=boxes, masks=masks, obj_to_img=obj_to_img,
                attribute=attribute, masks_shift=masks_shift, boxes_shift=boxes_shift,
                z=z, attribute_gt=attribute_gt, attribute_est=attribute_est)


def make_pos_weight(n_attr: int = NUM_ATTRIBUTES, seed: int = 7) -> np.ndarray:
    """Stand-in for train64.py:25-28: pos_weight[k] = (100000 - c_k) / c_k with
    synthetic counts c_k log-uniform in [200, 20000] (the real counts file is data, not code)."""
    rng = np.random.default_rng(seed)
    c = np.exp(rng.uniform(math.log(200.0), math.log(20000.0), size=n_attr))
    return ((100000.0 - c) / c).astype(np.float32)


def make_cooccurrence(n_classes: int = NUM_OBJECT_CLASSES, n_attr: int = NUM_ATTRIBUTES, seed: int = 11) -> np.ndarray:
    """Stand-in for matrix_obj_vs_att.pt (train64.py:83): (V, A) non-negative co-occurrence counts, sparse like the
    real table (most object classes see a handful of attributes)."""
    rng = np.random.default_rng(seed)
    m = rng.integers(1, 500, size=(n_classes, n_attr)).astype(np.float32)
    m *= rng.random((n_classes, n_attr)) < 0.3
    m[:, 0] += 1.0                                   # no all-zero row (random.choices would raise)
    return m


def shard(batch: Dict[str, np.ndarray], rank: int, world: int) -> Dict[str, np.ndarray]:
    """Contiguous image shard for data-parallel rank `rank`; obj_to_img renumbered from 0."""
    n = batch["imgs"].shape[0]
    assert n % world == 0, "global batch must divide evenly over ranks"
    per = n // world
    lo, hi = rank * per, (rank + 1) * per
    sel = (batch["obj_to_img"] >= lo) & (batch["obj_to_img"] < hi)
    out = {}
    for k, v in batch.items():
        if k == "imgs":
            out[k] = v[lo:hi]
        elif k == "obj_to_img":
            out[k] = v[sel] - lo
        else:
            out[k] = v[sel]
    return out
