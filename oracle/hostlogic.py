"""TEST INFRASTRUCTURE ONLY — CPU restatement of the host logic around the train step (never imported by the
product path).  Follows the reference line by line, slow loops and all:

  estimate_attributes      train64.py:156-166
  swap_attributes          train64.py:153,170-188
  imagenet_deprocess_batch data/utils.py:32-66 (torchvision Normalize = sub mean, div std, in fp32)

Parity note: train64.py cannot be imported here (tensorboardX/h5py/torchvision missing), so these are pinned by
restatement only; the draws use python's `random` exactly as the reference's call sites do.
"""
import math
import random

import torch


def estimate_attributes(att_logits, attribute):
    att_idx = attribute.sum(dim=1).nonzero().view(-1)                 # train64.py:144
    est = attribute.clone()
    att_mask = torch.zeros(attribute.shape[0]).scatter(0, att_idx, 1)
    max_idx = att_logits.argmax(1).float() * (1 - att_mask)
    annotated = set(int(v) for v in att_idx)
    for row in range(attribute.shape[0]):
        if row not in annotated:
            est[row, int(max_idx[row])] = 1
    return est


def swap_attributes(attribute, attribute_est, objs, obj_to_img, matrix, n_images, rng=random):
    """Mutates attribute / attribute_est like the loop body of train64.py:170-188."""
    attribute_gt = attribute.clone()
    n_attr = attribute.shape[1]
    for img_idx in range(math.floor(n_images / 3)):
        obj_indices = torch.nonzero(obj_to_img == img_idx).view(-1)
        limit = math.floor(len(obj_indices) / 2)
        for changed, obj_idx in enumerate(obj_indices):
            if changed >= limit:
                break
            obj = objs[obj_idx]
            old = torch.nonzero(attribute_gt[obj_idx]).view(-1)
            new = rng.choices(range(n_attr), matrix[obj].scatter(0, old, 0), k=rng.randrange(1, 3))
            attribute[obj_idx] = 0
            attribute[obj_idx] = attribute[obj_idx].scatter(0, torch.LongTensor(new), 1)
            attribute_est[obj_idx] = 0
            attribute_est[obj_idx] = attribute[obj_idx].scatter(0, torch.LongTensor(new), 1)
    return attribute, attribute_est


IMAGENET_MEAN = [0.485, 0.456, 0.406]
IMAGENET_STD = [0.229, 0.224, 0.225]


def _normalize(t, mean, std):
    mean = torch.as_tensor(mean, dtype=t.dtype)[:, None, None]
    std = torch.as_tensor(std, dtype=t.dtype)[:, None, None]
    return t.clone().sub_(mean).div_(std)


def imagenet_deprocess_batch(imgs, rescale=True):
    imgs = imgs.detach().cpu().clone()
    out = []
    for i in range(imgs.size(0)):
        x = _normalize(imgs[i], [0, 0, 0], [1.0 / s for s in IMAGENET_STD])
        x = _normalize(x, [-m for m in IMAGENET_MEAN], [1.0, 1.0, 1.0])
        if rescale:
            lo, hi = x.min(), x.max()
            x = x.sub(lo).div(hi - lo)
        out.append(x[None].mul(255).clamp(0, 255).byte())
    return torch.cat(out, dim=0)


# ---- N3: per-object layout tensors from boxes (data/vg_custom_mask.py:136-158), python floats and python round like the reference
def layout_from_boxes(boxes, R):
    O = boxes.shape[0]
    masks = torch.zeros(O, 1, R, R)
    masks_shift = torch.zeros(O, 1, R, R)
    boxes_shift = torch.FloatTensor([[0, 0, 1, 1]]).repeat(O, 1)
    for i in range(O):
        x0, y0, x1, y1 = (float(v) for v in boxes[i])
        masks[i, :, round(y0 * R):round(y1 * R), round(x0 * R):round(x1 * R)] = 1          # :136
        width = x1 - x0
        x0_shift, x1_shift = x0, x1
        if width < 0.5:                                                                    # :144
            border_dist_left = x0
            border_dist_right = 1 - x1
            if border_dist_left > border_dist_right:
                shift = border_dist_left * 0.8
                x0_shift, x1_shift = x0 - shift, x1 - shift
            elif border_dist_right > border_dist_left:
                shift = border_dist_right * 0.8
                x0_shift, x1_shift = x0 + shift, x1 + shift
        masks_shift[i, :, round(y0 * R):round(y1 * R), round(x0_shift * R):round(x1_shift * R)] = 1   # :157
        boxes_shift[i] = torch.FloatTensor([x0_shift, y0, x1_shift, y1])                              # :158
    return boxes_shift, masks, masks_shift


# ---- N2: the attribute logic of the inference loop (test64.py:143-150, 160-167, 179-184) on CPU tensors
def edit_attribute_rows(attribute, attribute_est, tgt, remove=(2, 8, 0, 94, 90, 95, 96, 34, 25, 70, 58, 104)):
    attribute, attribute_est = attribute.clone(), attribute_est.clone()
    for idx in range(attribute.shape[0]):
        attribute[idx, list(remove)] = 0
        attribute[idx, tgt] = 1
        attribute_est[idx, list(remove)] = 0
        attribute_est[idx, tgt] = 1
    return attribute, attribute_est


def edit_success(logits_rand, logits_rand_edit, tgt):
    max_idx = logits_rand.topk(5)[1]
    changed = [i for i in range(logits_rand.shape[0]) if tgt not in max_idx[i]]
    max_idx_y = logits_rand_edit.topk(3)[1]
    return changed, [i for i in changed if tgt in max_idx_y[i]]
