"""Inference / attribute-editing loop body of the reference's test64.py:114-198 (test128.py is the same with the 128 px
modules) on the eval-mode HIP kernels — SURVEY.md §8f N2.

Per batch the reference (1) estimates attributes of un-annotated objects with D_att (:126-135), (2) generates images,
(3) scores the attribute classifier on the generated crops of annotated objects (sigmoid > 0.9, :143-150), (4) rewrites one
attribute group of EVERY object (remove a list of colour attributes, set the target, :160-167), (5) generates again with a
fresh z, (6) counts an edit as successful when the target attribute is not in the top-5 of D_att on the first crops but is in
the top-3 on the edited ones (:180-184), and (7) de-normalises the images to bytes (:153-155,173-176).  Here every tensor
operation is a libagl.so launch; the only host work is the z draws (the reference's torch.randn on the CPU) and the final copies.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import functional as F
from . import lib as L
from .hostlogic import IMAGENET_MEAN, IMAGENET_STD

COLOUR_ATTRIBUTES = (2, 8, 0, 94, 90, 95, 96, 34, 25, 70, 58, 104)     # test64.py:163 "remove other color"


def _deprocess(x):
    import numpy as np
    inv_std = [float(np.float32(1.0 / s)) for s in IMAGENET_STD]
    mean = [float(np.float32(m)) for m in IMAGENET_MEAN]
    return L.deprocess_u8(x.detach().float().contiguous(), inv_std, mean, True)


def edit_attributes_batch(netG, netD_att, batch: Dict[str, torch.Tensor], *, tgt: int = 95,
                          remove: Sequence[int] = COLOUR_ATTRIBUTES, z: Optional[torch.Tensor] = None,
                          z_edit: Optional[torch.Tensor] = None, eps: Optional[Sequence[torch.Tensor]] = None,
                          eps_edit: Optional[Sequence[torch.Tensor]] = None, threshold: float = 0.9) -> Dict[str, torch.Tensor]:
    """One iteration of the loop of test64.py:114-198.  `batch`: device tensors imgs, objs, boxes, masks, attribute, masks_shift,
    boxes_shift and the CPU obj_to_img.  z / z_edit (O, z_dim) and the eps triples pin the random draws (default: torch.randn on
    the CPU like the reference).  Returns device tensors; nothing is written to disk.

    Module modes follow the reference: test64.py:114 puts ONLY netG in eval mode; netD_att keeps the mode it has (the reference
    never calls netD_att.eval(), so there its spectral norm runs one power iteration in each of its FOUR forward calls per batch:
    :129, :146, :180, :183 — this function makes the same four calls in the same order, so u/v advance identically).  Call
    netD_att.eval() first for a state-free loop.  netG's training flag is restored on return."""
    dev = batch["imgs"].device
    objs, attribute = batch["objs"], batch["attribute"].contiguous()
    O = objs.shape[0]
    zdim = netG.z_dim if hasattr(netG, "z_dim") else 64
    z = (torch.randn(O, zdim) if z is None else z).to(dev)
    z_edit = (torch.randn(O, zdim) if z_edit is None else z_edit).to(dev)
    o2i = batch["obj_to_img"]
    g_was_training = netG.training
    try:
        return _edit(netG, netD_att, batch, dev, objs, attribute, o2i, z, z_edit, eps, eps_edit, tgt, remove, threshold)
    finally:
        netG.train(g_was_training)


def _edit(netG, netD_att, batch, dev, objs, attribute, o2i, z, z_edit, eps, eps_edit, tgt, remove, threshold):
    with torch.no_grad():
        netG.eval()
        # (1) attribute estimate (:126-135)
        crops_input = F.crop_boxes(batch["imgs"], batch["boxes"], F.L.box_map_to_device(o2i, dev), netG.obj_size)
        attribute_est = L.attr_estimate(netD_att(crops_input), attribute)
        # (2) generate (:138-139)
        out = netG(batch["imgs"], objs, batch["boxes"], batch["masks"], o2i, z, attribute, batch["masks_shift"], batch["boxes_shift"],
                   attribute_est, eps=eps)
        crops_rand, img_rec, img_rand, img_shift = out[2], out[4], out[5], out[6]
        # (3) attribute classifier on the generated crops (:142-150).  The reference passes the annotated rows only; a row's
        # logits do not depend on the other rows, so all rows are scored here and the caller masks with `annotated`
        pred = L.sigmoid_threshold(netD_att(crops_rand), threshold)
        # (4) attribute modification for every object (:160-167)
        cols = torch.tensor(list(remove), dtype=torch.int32, device=dev)
        attribute_new = L.attr_edit_(attribute.clone(), cols, tgt)
        attribute_est_new = L.attr_edit_(attribute_est.clone(), cols, tgt)
        # (5) generate the edited images with a fresh z (:170-171)
        out_y = netG(batch["imgs"], objs, batch["boxes"], batch["masks"], o2i, z_edit, attribute_new, batch["masks_shift"],
                     batch["boxes_shift"], attribute_est_new, eps=eps_edit)
        crops_rand_y, img_rec_y, img_rand_y, img_shift_y = out_y[2], out_y[4], out_y[5], out_y[6]
        # (6) success statistics (:179-184): the reference scores crops_rand again here (:180, its third D_att call)
        logits_rand = netD_att(crops_rand)
        changed = L.topk_contains(logits_rand, 5, tgt) == 0                 # target not yet among the top-5
        success = changed & (L.topk_contains(netD_att(crops_rand_y), 3, tgt) != 0)
        # (7) bytes (:153-155, :173-176)
        images = {k: _deprocess(v) for k, v in (("real", batch["imgs"]), ("rec", img_rec), ("rand", img_rand), ("shift", img_shift),
                                                 ("rec_edit", img_rec_y), ("rand_edit", img_rand_y), ("shift_edit", img_shift_y))}
    return {"attribute_est": attribute_est, "pred": pred, "annotated": attribute.sum(dim=1) != 0, "attribute_edit": attribute_new,
            "attribute_est_edit": attribute_est_new, "changed": changed, "success": success, "logits_rand": logits_rand,
            "images": images, "img_rand": img_rand, "img_rand_edit": img_rand_y}
