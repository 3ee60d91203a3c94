"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/graph.py header).

CPU restatement of one G+D training iteration of the reference's train64.py / train128.py
(they differ only in the generator module, AttributeDiscriminator128 and default sizes).
Follows train64.py:160-161 (pre-step D_att forward on real crops), :191-262 (D step) and
:280-370 (G step), with the lambda defaults of :439-446 and four Adam(2e-4, (.5,.999))
optimisers (:111-114).

Deliberate, documented deviations:
  * the argmax->attribute_est python loop (:162-166) and the random attribute swap
    (:170-188) are host data preparation, not arithmetic of the path; the batch supplies
    `attribute` (post-swap), `attribute_gt` (pre-swap) and `attribute_est` directly.
    The D_att forward that feeds the loop is still executed (it advances SN u/v).
  * RNG is pinned: `z` and the 2x3 crop-encoder `eps` draws are inputs.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import graph as G

BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v")

LAMBDAS = dict(img_adv=1.0, obj_adv=1.0, obj_cls=1.0, z_rec=8.0, img_rec=1.0, kl=0.01, att_cls=2.0)   # train64.py:439-446
LR, BETAS, ADAM_EPS = 2e-4, (0.5, 0.999), 1e-8                                                        # train64.py:434,111-114


def is_buffer(name: str) -> bool:
    return name.endswith(BUFFER_SUFFIXES)


def as_params(state: Dict[str, torch.Tensor]) -> G.Params:
    """Clone a state_dict into leaf tensors (parameters require grad, buffers do not)."""
    out = {}
    for k, v in state.items():
        t = v.detach().clone().cpu()
        if not is_buffer(k) and t.is_floating_point():
            t.requires_grad_(True)
        out[k] = t
    return out


def leaves(P: G.Params):
    return [v for k, v in P.items() if v.requires_grad]


class OracleBackend:
    """The four networks as flat parameter dictionaries driven through oracle.graph."""

    def __init__(self, g_state, dimg_state, dobj_state, datt_state, *, res128: bool, obj_size: int):
        self.Pg, self.Pi, self.Po, self.Pa = (as_params(s) for s in (g_state, dimg_state, dobj_state, datt_state))
        self.res128, self.obj_size = res128, obj_size
        mk = lambda P: torch.optim.Adam(leaves(P), LR, BETAS, eps=ADAM_EPS)
        self.opt_g, self.opt_i, self.opt_o, self.opt_a = mk(self.Pg), mk(self.Pi), mk(self.Po), mk(self.Pa)

    def crop(self, feats, boxes, o2i):
        return G.crop_boxes(feats, boxes, o2i, self.obj_size)

    def gen(self, b, eps):
        return G.generator(self.Pg, b["imgs"], b["objs"], b["boxes"], b["masks"], b["obj_to_img"], b["z"],
                           b["attribute"], b["masks_shift"], b["boxes_shift"], b["attribute_est"],
                           obj_size=self.obj_size, res128=self.res128, train=True, eps=eps)

    def d_img(self, x):
        return G.image_discriminator(self.Pi, x, True)

    def d_obj(self, x, objs):
        return G.object_discriminator(self.Po, x, True)

    def d_att(self, x):
        return G.attribute_discriminator(self.Pa, x, True, self.res128)

    @staticmethod
    def _zero(P):
        for v in P.values():
            v.grad = None

    def zero_d(self):
        for P in (self.Pi, self.Po, self.Pa):
            self._zero(P)

    def zero_g(self):
        self._zero(self.Pg)

    def step_d(self):
        self.opt_i.step(); self.opt_o.step(); self.opt_a.step()

    def step_g(self):
        self.opt_g.step()

    def states(self):
        return {"G": self.Pg, "D_img": self.Pi, "D_obj": self.Po, "D_att": self.Pa}


def run_step(be, b: Dict[str, torch.Tensor], pos_weight: torch.Tensor,
             eps_d: Optional[Sequence[torch.Tensor]] = None, eps_g: Optional[Sequence[torch.Tensor]] = None,
             lambdas: Optional[dict] = None, on_d_backward=None, on_g_backward=None):
    """One training iteration on backend `be` (OracleBackend here; make_golden.py drives the
    imported reference nn.Modules through the very same function).  Returns (losses, G outputs)."""
    lam = dict(LAMBDAS, **(lambdas or {}))
    objs = b["objs"]
    N = b["imgs"].shape[0]
    n_swap = math.floor(N / 3)                                    # train64.py:170
    losses: Dict[str, torch.Tensor] = {}

    # ---- pre-step: attribute estimate forward (train64.py:160-161)
    with torch.no_grad():
        be.d_att(be.crop(b["imgs"], b["boxes"], b["obj_to_img"]))

    # ---- D step (train64.py:191-262)
    out = be.gen(b, eps_d)
    crops_input, crops_rec, crops_rand, crops_shift, img_rec, img_rand, img_shift = (t.detach() for t in out[:7])
    l_i_rec = G.bce_logits(be.d_img(img_rec), 0.0)
    l_i_rand = G.bce_logits(be.d_img(img_rand), 0.0)
    l_i_shift = G.bce_logits(be.d_img(img_shift), 0.0)
    d_img_fake = 0.4 * l_i_rec + 0.4 * l_i_rand + 0.2 * l_i_shift
    d_img_real = G.bce_logits(be.d_img(b["imgs"]), 1.0)
    l_o_rec = G.bce_logits(be.d_obj(crops_rec, objs)[0], 0.0)
    l_o_rand = G.bce_logits(be.d_obj(crops_rand, objs)[0], 0.0)
    l_o_shift = G.bce_logits(be.d_obj(crops_shift, objs)[0], 0.0)
    d_obj_fake = 0.4 * l_o_rec + 0.4 * l_o_rand + 0.2 * l_o_shift
    src, cls = be.d_obj(crops_input, objs)
    d_obj_real = G.bce_logits(src, 1.0)
    d_obj_cls = F.cross_entropy(cls, objs)
    att = be.d_att(crops_input)
    rows = b["attribute_gt"].sum(dim=1).nonzero().view(-1)        # :241
    d_att = F.binary_cross_entropy_with_logits(att.index_select(0, rows), b["attribute_gt"].index_select(0, rows),
                                               pos_weight=pos_weight)
    d_loss = (lam["img_adv"] * (d_img_fake + d_img_real) + lam["obj_adv"] * (d_obj_fake + d_obj_real)
              + lam["obj_cls"] * d_obj_cls + lam["att_cls"] * d_att)
    be.zero_d()
    d_loss.backward()
    if on_d_backward is not None:
        on_d_backward(be)
    be.step_d()
    losses.update({"D/loss": d_loss, "D/image_adv_loss_real": d_img_real, "D/image_adv_loss_fake": d_img_fake,
                   "D/object_adv_loss_real": d_obj_real, "D/object_adv_loss_fake": d_obj_fake,
                   "D/object_cls_loss_real": d_obj_cls, "D/object_att_cls_loss": d_att})

    # ---- G step (train64.py:280-370)
    out = be.gen(b, eps_g)
    (crops_input, crops_rec, crops_rand, crops_shift, img_rec, img_rand, img_shift,
     mu, logvar, z_rand_rec, z_rand_shift) = out
    keep = torch.ones(N)
    keep[:n_swap] = 0                                             # :284
    g_img_rec = (keep * (img_rec - b["imgs"]).abs().reshape(N, -1).mean(1)).sum() / (N - n_swap)
    g_z_rec = 0.5 * (z_rand_rec - b["z"]).abs().mean() + 0.5 * (z_rand_shift - b["z"]).abs().mean()
    g_kl = G.kl_sum(mu, logvar)
    g_img_adv = (0.4 * G.bce_logits(be.d_img(img_rec), 1.0) + 0.4 * G.bce_logits(be.d_img(img_rand), 1.0)
                 + 0.2 * G.bce_logits(be.d_img(img_shift), 1.0))
    rows = b["attribute"].sum(dim=1).nonzero().view(-1)           # :323
    tgt = b["attribute"].index_select(0, rows)
    adv, clsl, attl = [], [], []
    for crops in (crops_rec, crops_rand, crops_shift):            # :316-349 (D_obj then D_att per crop set)
        s, c = be.d_obj(crops, objs)
        adv.append(G.bce_logits(s, 1.0))
        clsl.append(F.cross_entropy(c, objs))
        a = be.d_att(crops)
        attl.append(F.binary_cross_entropy_with_logits(a.index_select(0, rows), tgt, pos_weight=pos_weight))
    mix = lambda t: 0.4 * t[0] + 0.4 * t[1] + 0.2 * t[2]
    g_obj_adv, g_obj_cls, g_att = mix(adv), mix(clsl), mix(attl)
    g_loss = (lam["img_rec"] * g_img_rec + lam["z_rec"] * g_z_rec + lam["img_adv"] * g_img_adv
              + lam["obj_adv"] * g_obj_adv + lam["obj_cls"] * g_obj_cls + lam["att_cls"] * g_att
              + lam["kl"] * g_kl)
    be.zero_g()
    g_loss.backward()          # also deposits grads in the D nets; the reference zeroes them before
    if on_g_backward is not None:
        on_g_backward(be)
    be.step_g()                # its next D backward (:254-256) and never consumes them
    losses.update({"G/loss": g_loss, "G/image_adv_loss": g_img_adv, "G/object_adv_loss": g_obj_adv,
                   "G/object_cls_loss": g_obj_cls, "G/rec_img": g_img_rec, "G/rec_z": g_z_rec, "G/kl": g_kl,
                   "G/object_att_cls_loss": g_att})
    return {k: float(v.detach()) for k, v in losses.items()}, [t.detach() for t in out]
