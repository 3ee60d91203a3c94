"""Drop-in for the differentiable crop of the reference's models/bilinear.py (crop_bbox_batch :26,
crop_bbox_batch_cudnn :67, crop_bbox :107): one gather kernel indexed by bbox_to_feats instead of the
reference's per-image expand + cat + grid_sample + inverse permutation."""
from . import _bootstrap  # noqa: F401
import torch
from agl import functional as _F


def crop_bbox_batch(feats, bbox, bbox_to_feats, HH, WW=None, backend='cudnn', align_corners=False):
    """crops[b] = bilinear resample of feats[bbox_to_feats[b]] over bbox[b] = [x0,y0,x1,y1] in [0,1]
    (zero padding outside the map).  `bbox_to_feats` may live on the CPU, as in the reference's loop."""
    if backend not in ('cudnn', 'jj'):
        raise ValueError("unknown backend %r" % (backend,))
    if backend == 'jj':
        raise NotImplementedError("the 'jj' backend is never used by the reference training path")
    assert bbox.size(1) == 4 and bbox.size(0) == bbox_to_feats.size(0)
    if not bbox_to_feats.is_cuda and bbox_to_feats.numel():        # host-resident index (the reference's loop): validate for free
        lo, hi = int(bbox_to_feats.min()), int(bbox_to_feats.max())
        if lo < 0 or hi >= feats.size(0):
            raise IndexError("crop_bbox_batch: bbox_to_feats must lie in [0, %d), got [%d, %d]" % (feats.size(0), lo, hi))
    return _F.crop_boxes(feats, bbox, _F.L.box_map_to_device(bbox_to_feats, feats.device), HH, WW, align_corners)


crop_bbox_batch_cudnn = crop_bbox_batch


def crop_bbox(feats, bbox, HH, WW=None, backend='cudnn', align_corners=False):
    """Per-map crop: crops[i] from feats[i] (reference :107)."""
    assert bbox.size(0) == feats.size(0) and bbox.size(1) == 4
    idx = torch.arange(feats.size(0), device=feats.device)
    idx._agl_sorted = True      # (one box per map, in order: the fixed-order backward applies)
    return _F.crop_boxes(feats, bbox, idx, HH, WW, align_corners)
