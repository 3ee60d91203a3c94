"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (functional PyTorch, fp32) of the reference's G+D module graph.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file; the shipped path (`attribute-guided-image-generation-from-layout_amd/`)
runs on hand-written HIP kernels and raises if its library is missing.

Why PyTorch-CPU and not C/numpy: the reference contains no arithmetic of its own —
every number on the hot path is produced by third-party PyTorch ops (SURVEY.md §8c,
"Where the arithmetic really lives").  This file therefore restates the reference's
*graph* (which op, on which operand, in which order, with which state side effects)
as plain functions over one flat ``{state_dict key: tensor}`` dictionary, and calls
the same torch CPU ops at the leaves.  The two non-trivial leaves are restated in
closed form in this file: the bilinear crop (`crop_boxes`, models/bilinear.py:26-136)
and the spectral-norm power iteration (`sn_weight`, torch.nn.utils.spectral_norm).

Parity pin: `oracle/make_golden.py` imports the real reference from /root/reference
in the build container, checks this restatement against it (max |diff| printed, must
be <= 1e-6 abs on outputs, grads and post-step state) and writes tests/golden/*.npz.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

BN_MOMENTUM = 0.1
BN_EPS = 1e-5
SN_EPS = 1e-12


# --------------------------------------------------------------------------- optional operand rounding (bf16-mode comparisons)
# OPERAND_ROUND = None: every convolution below is the plain torch op (the fp32 path the reference fixtures pin — untouched).
# OPERAND_ROUND = a callable r(t) -> t (e.g. `lambda t: t.to(torch.bfloat16).to(torch.float32)`): the operands of every
# convolution with >= OPERAND_ROUND_MIN_CIN input channels are rounded where the HIP bf16 mode rounds them — x and w in the forward,
# dy and w in the input gradient, dy and x in the weight gradient — with fp32 accumulation.  This is NOT the reference's arithmetic:
# it is the yardstick for "are the bf16-mode kernels doing bf16 arithmetic correctly", next to the fp32 oracle that says what the
# mode costs (tests/test_model_gpu.py).  Layers with fewer input channels (RGB-side layers) run in exact fp32 in the HIP path too.
OPERAND_ROUND = None
OPERAND_ROUND_MIN_CIN = 16


class _RoundedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, transposed):
        r = OPERAND_ROUND
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, transposed, r)
        if transposed:
            return F.conv_transpose2d(r(x), r(w), None, stride=stride, padding=padding)
        return F.conv2d(r(x), r(w), None, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, transposed, r = ctx.cfg
        dyr = r(dy)
        if transposed:      # y = convT(x, w): dx = conv(dy, w); dw[ci][co] = the weight gradient of that convolution with the roles swapped
            dx = F.conv2d(dyr, r(w), None, stride=stride, padding=padding)
            dw = torch.nn.grad.conv2d_weight(dyr, w.shape, r(x), stride=stride, padding=padding)
        else:
            dx = torch.nn.grad.conv2d_input(x.shape, r(w), dyr, stride=stride, padding=padding)
            dw = torch.nn.grad.conv2d_weight(r(x), w.shape, dyr, stride=stride, padding=padding)
        return dx, dw, None, None, None


def _conv2d(x, w, b=None, stride=1, padding=0):
    if OPERAND_ROUND is None or w.shape[1] < OPERAND_ROUND_MIN_CIN:
        return F.conv2d(x, w, b, stride=stride, padding=padding)
    y = _RoundedConv.apply(x, w, stride, padding, False)
    return y if b is None else y + b.view(1, -1, 1, 1)


def _conv_transpose2d(x, w, b=None, stride=1, padding=0):
    if OPERAND_ROUND is None or w.shape[0] < OPERAND_ROUND_MIN_CIN:
        return F.conv_transpose2d(x, w, b, stride=stride, padding=padding)
    y = _RoundedConv.apply(x, w, stride, padding, True)
    return y if b is None else y + b.view(1, -1, 1, 1)


# --------------------------------------------------------------------------- crop
def _linspace_pair(steps: int, like: torch.Tensor):
    # models/bilinear.py:272-275 — two torch.linspace weight ramps
    w_start = torch.linspace(1, 0, steps=steps).to(like)
    w_end = torch.linspace(0, 1, steps=steps).to(like)
    return w_start, w_end


def crop_boxes(feats, boxes, box_to_img, HH, WW=None, align_corners=False):
    """models/bilinear.py:26 -> :67 -> :107 -> F.grid_sample (:136).

    Output row b is the bilinear resample of feats[box_to_img[b]] over boxes[b]
    ([x0,y0,x1,y1] in [0,1]).  The reference gathers per image and un-permutes at the
    end (:99-104); the net effect for any box_to_img is the direct per-box gather done
    here.
    """
    if WW is None:
        WW = HH
    idx = box_to_img.to(feats.device).long()
    src = feats.index_select(0, idx)
    b = 2 * boxes - 1                                    # :127
    x0, y0, x1, y1 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ws, we = _linspace_pair(WW, b)
    hs, he = _linspace_pair(HH, b)
    X = ws[None, :] * x0[:, None] + we[None, :] * x1[:, None]      # :277-282
    Y = hs[None, :] * y0[:, None] + he[None, :] * y1[:, None]
    B = boxes.shape[0]
    grid = torch.stack([X[:, None, :].expand(B, HH, WW), Y[:, :, None].expand(B, HH, WW)], dim=3)
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)


# --------------------------------------------------------------------------- norms
def _bn(P: Params, pre: str, x, train: bool, affine: bool):
    """nn.BatchNorm{1,2}d: batch statistics in training (biased var to normalise,
    unbiased into running_var, momentum .1, eps 1e-5) — torch semantics."""
    w = P[pre + "weight"] if affine else None
    b = P[pre + "bias"] if affine else None
    if train:
        with torch.no_grad():
            P[pre + "num_batches_tracked"] += 1
    return F.batch_norm(x, P[pre + "running_mean"], P[pre + "running_var"], w, b, train, BN_MOMENTUM, BN_EPS)


def cond_bn(P: Params, pre: str, x, labels, train: bool):
    """models/generator_obj_att.py:31-44 ConditionalBatchNorm2d."""
    C = x.shape[1]
    xh = _bn(P, pre + "bn.", x, train, affine=False)
    gb = F.embedding(labels, P[pre + "embed.weight"])
    return gb[:, :C, None, None] * xh + gb[:, C:, None, None]


def spade(P: Params, pre: str, x, seg, train: bool):
    """models/spade/networks/normalization.py:94-108 (param_free_norm = BatchNorm2d, :77-78)."""
    xh = _bn(P, pre + "param_free_norm.", x, train, affine=False)
    seg = F.interpolate(seg, size=x.shape[2:], mode="nearest")
    a = F.relu(_conv2d(seg, P[pre + "mlp_shared.0.weight"], P[pre + "mlp_shared.0.bias"], padding=1))
    gamma = _conv2d(a, P[pre + "mlp_gamma.weight"], P[pre + "mlp_gamma.bias"], padding=1)
    beta = _conv2d(a, P[pre + "mlp_beta.weight"], P[pre + "mlp_beta.bias"], padding=1)
    return xh * (1 + gamma) + beta


# --------------------------------------------------------------------------- generator pieces
def crop_encoder(P: Params, pre: str, crops, labels, train: bool, eps: Optional[torch.Tensor]):
    """models/generator_obj_att.py:395-422 CropEncoder.forward (class-conditional branch)."""
    x = crops
    for conv, bn, stride, pad in (("c1", "bn1", 1, 3), ("c2", "bn2", 2, 1), ("c3", "bn3", 2, 1),
                                  ("c4", "bn4", 2, 1), ("conv5", "bn5", 2, 1)):
        x = _conv2d(x, P[pre + conv + ".weight"], None, stride=stride, padding=pad)
        x = F.relu(cond_bn(P, pre + bn + ".", x, labels, train))
    x = x.mean(dim=(2, 3))                                # AdaptiveAvgPool2d(1) + view
    mu = F.linear(x, P[pre + "fc_mu.weight"], P[pre + "fc_mu.bias"])
    logvar = F.linear(x, P[pre + "fc_logvar.weight"], P[pre + "fc_logvar.bias"])
    std = torch.exp(0.5 * logvar)
    if eps is None:
        eps = torch.randn(std.shape[0], std.shape[1])     # :10-15, :419 (CPU RNG)
    z = eps.to(std) * std + mu
    return z, mu, logvar


def attribute_encoder(P: Params, pre: str, labels, attr, train: bool):
    """models/generator_obj_att.py:588-600."""
    a = torch.cat((F.embedding(labels, P[pre + "embedding.weight"]), attr), dim=1)
    a = F.linear(a, P[pre + "c0.weight"], P[pre + "c0.bias"])
    a = F.relu(_bn(P, pre + "bn0.", a, train, affine=True))
    a = F.linear(a, P[pre + "c1.weight"], P[pre + "c1.bias"])
    a = F.relu(_bn(P, pre + "bn1.", a, train, affine=True))
    return F.linear(a, P[pre + "c2.weight"], P[pre + "c2.bias"])


def conv_lstm_fuse(P: Params, pre: str, feats, box_to_img, hidden: Sequence[int] = (128, 64, 64)):
    """models/generator_obj_att.py:271-346 LayoutConvLSTM.forward with ConvLSTMCell :99-114.

    Objects of one image are consecutive in `feats`; each image's run is one sequence.
    Per image, per layer, per time-step: cc = conv5x5(cat[x_t, h]); gates split in the
    order i, f, o, g (:105); zero initial state (:116-118).  Returns the last h of the
    last layer for every image, concatenated (:341-344).
    """
    ids = [int(v) for v in box_to_img.tolist()]
    runs: List[Tuple[int, int]] = []
    start = 0
    for k in range(1, len(ids) + 1):                      # :286-304 run-length split
        if k == len(ids) or ids[k] != ids[k - 1]:
            runs.append((start, k))
            start = k
    outs = []
    S = feats.shape[2:]
    for (a, b) in runs:
        seq = [feats[t:t + 1] for t in range(a, b)]
        for li, hid in enumerate(hidden):
            w = P[f"{pre}cell_list.{li}.conv.weight"]
            bia = P[f"{pre}cell_list.{li}.conv.bias"]
            h = feats.new_zeros(1, hid, *S)
            c = feats.new_zeros(1, hid, *S)
            nxt = []
            for x_t in seq:
                cc = _conv2d(torch.cat([x_t, h], dim=1), w, bia, padding=2)
                gi, gf, go, gg = torch.split(cc, hid, dim=1)
                c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
                h = torch.sigmoid(go) * torch.tanh(c)
                nxt.append(h)
            seq = nxt
        outs.append(seq[-1])
    return torch.cat(outs, dim=0)


def layout_encoder(P: Params, pre: str, obj_vec, masks, box_to_img, z, labels, train: bool, pool_to_8: bool):
    """models/generator_obj_att.py:487-513 (64 px) / models/generator_obj_att128.py:489-511 (128 px,
    adds AdaptiveAvgPool2d(8) after bn4, :486,505)."""
    v = torch.cat((obj_vec, z), dim=1)
    h = v[:, :, None, None] * masks
    h = _conv2d(h, P[pre + "c0.weight"], None, stride=1, padding=1)      # k1 p1 -> R+2
    h = F.relu(cond_bn(P, pre + "bn1.", h, labels, train))
    h = _conv2d(h, P[pre + "c2.weight"], None, stride=2, padding=1)
    h = F.relu(cond_bn(P, pre + "bn2.", h, labels, train))
    h = _conv2d(h, P[pre + "c3.weight"], None, stride=2, padding=1)
    h = F.relu(cond_bn(P, pre + "bn3.", h, labels, train))
    h = _conv2d(h, P[pre + "c4.weight"], None, stride=2, padding=1)
    h = cond_bn(P, pre + "bn4.", h, labels, train)                        # no ReLU after bn4
    if pool_to_8:
        h = F.adaptive_avg_pool2d(h, 8)
    h = conv_lstm_fuse(P, pre + "clstm.", h, box_to_img)
    for r in range(6):                                                    # ResidualBlock :47-60
        rp = f"{pre}residual.{r}.main."
        t = _conv2d(h, P[rp + "0.weight"], None, padding=1)
        t = F.relu(_bn(P, rp + "1.", t, train, affine=True))
        t = _conv2d(t, P[rp + "3.weight"], None, padding=1)
        t = _bn(P, rp + "4.", t, train, affine=True)
        h = h + t
    return h


def global_encoder(P: Params, pre: str, h, train: bool):
    """models/generator_obj_att.py:437-446."""
    h = _conv2d(h, P[pre + "c1.weight"], None, stride=2, padding=1)
    h = F.relu(_bn(P, pre + "bn1.", h, train, affine=True))
    h = _conv2d(h, P[pre + "c2.weight"], None, stride=2, padding=1)
    return h.sum(dim=(2, 3))


def decoder(P: Params, pre: str, hidden, glob, train: bool, res128: bool):
    """models/generator_obj_att.py:546-572 ; 128: models/generator_obj_att128.py:560-604."""
    seg = hidden
    h = torch.cat((hidden, glob[:, :, None, None].expand(-1, -1, 8, 8)), dim=1)
    h = _conv2d(h, P[pre + "c0_new.weight"], None, padding=1)
    h = F.relu(spade(P, pre + "spade_0.", h, seg, train))
    h = _conv_transpose2d(h, P[pre + "dc1.weight"], None, stride=2, padding=1)
    h = F.relu(spade(P, pre + "spade_1.", h, seg, train))
    h = _conv_transpose2d(h, P[pre + "dc2.weight"], None, stride=2, padding=1)
    h = F.relu(spade(P, pre + "spade_2.", h, seg, train))
    h = _conv_transpose2d(h, P[pre + "dc3.weight"], None, stride=2, padding=1)
    h = F.relu(spade(P, pre + "spade_3.", h, seg, train))
    h = _conv2d(h, P[pre + "c4.weight"], P[pre + "c4.bias"], padding=3)
    if not res128:
        return h
    h = F.interpolate(h, scale_factor=2, mode="nearest")
    h = _conv2d(h, P[pre + "c5.weight"], None, padding=3)
    h = F.relu(spade(P, pre + "spade_4.", h, seg, train))
    h = _conv2d(h, P[pre + "c6.weight"], None, padding=2)
    h = F.relu(spade(P, pre + "spade_5.", h, seg, train))
    return _conv2d(h, P[pre + "c7.weight"], P[pre + "c7.bias"], padding=3)


def generator(P: Params, imgs, objs, boxes, masks, obj_to_img, z_rand, attribute, masks_shift,
              boxes_shift, attribute_est, *, obj_size: int, res128: bool, train: bool = True,
              eps: Optional[Sequence[torch.Tensor]] = None):
    """models/generator_obj_att.py:618-647 (128: models/generator_obj_att128.py:650-679).

    `eps` optionally pins the three CPU randn draws of the crop encoder (in call order).
    """
    e = list(eps) if eps is not None else [None, None, None]
    crops_input = crop_boxes(imgs, boxes, obj_to_img, obj_size)
    z_rec, mu, logvar = crop_encoder(P, "crop_encoder.", crops_input, objs, train, e[0])
    oa = attribute_encoder(P, "attribute_encoder.", objs, attribute, train)
    oa_est = attribute_encoder(P, "attribute_encoder.", objs, attribute_est, train)
    le = lambda a, m, z: layout_encoder(P, "layout_encoder.", a, m, obj_to_img, z, objs, train, res128)
    h_rec = le(oa_est, masks, z_rec)
    h_rand = le(oa, masks, z_rand)
    h_shift = le(oa, masks_shift, z_rand)
    g_rec = global_encoder(P, "global_encoder.", h_rec, train)
    g_rand = global_encoder(P, "global_encoder.", h_rand, train)
    g_shift = global_encoder(P, "global_encoder.", h_shift, train)
    img_rec = decoder(P, "decoder.", h_rec, g_rec, train, res128)
    img_rand = decoder(P, "decoder.", h_rand, g_rand, train, res128)
    img_shift = decoder(P, "decoder.", h_shift, g_shift, train, res128)
    crops_rand = crop_boxes(img_rand, boxes, obj_to_img, obj_size)
    _, z_rand_rec, _ = crop_encoder(P, "crop_encoder.", crops_rand, objs, train, e[1])
    crops_input_rec = crop_boxes(img_rec, boxes, obj_to_img, obj_size)
    crops_shift = crop_boxes(img_shift, boxes_shift, obj_to_img, obj_size)
    _, z_rand_shift, _ = crop_encoder(P, "crop_encoder.", crops_shift, objs, train, e[2])
    return (crops_input, crops_input_rec, crops_rand, crops_shift, img_rec, img_rand, img_shift,
            mu, logvar, z_rand_rec, z_rand_shift)


# --------------------------------------------------------------------------- discriminators
def sn_weight(P: Params, pre: str, train: bool):
    """torch.nn.utils.spectral_norm as applied by models/discriminator.py:15-22 (add_sn):
    one power iteration per *training* forward, in-place on the u/v buffers, no grad;
    sigma = u^T W v ; W_sn = W / sigma (grad flows through W in numerator and sigma)."""
    w = P[pre + "weight_orig"]
    u = P[pre + "weight_u"]
    v = P[pre + "weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if train:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=SN_EPS))
    sigma = torch.dot(u.clone(), torch.mv(wm, v.clone()))
    return w / sigma


def _sn_conv(P, pre, x, train, padding):
    return _conv2d(x, sn_weight(P, pre, train), P[pre + "bias"], padding=padding)


def d_first_block(P: Params, pre: str, x, down: bool, train: bool):
    """models/discriminator.py:29-60 OptimizedBlock."""
    h = _sn_conv(P, pre + "resi.0.", x, train, 1)
    h = _sn_conv(P, pre + "resi.2.", F.relu(h), train, 1)
    if down:
        h = F.avg_pool2d(h, 2)
    s = F.avg_pool2d(x, 2) if down else x
    return h + _sn_conv(P, pre + "sc.", s, train, 0)


def d_res_block(P: Params, pre: str, x, train: bool):
    """models/discriminator.py:63-99 ResidualBlock(downsample=True).  `forward` evaluates
    residual(x) first and its leading ReLU is in-place (:71), so the shortcut conv reads
    relu(x) — restated explicitly here."""
    xr = F.relu(x)
    h = _sn_conv(P, pre + "resi.1.", xr, train, 1)
    h = _sn_conv(P, pre + "resi.3.", F.relu(h), train, 1)
    h = F.avg_pool2d(h, 2)
    s = F.avg_pool2d(_sn_conv(P, pre + "sc.", xr, train, 0), 2)
    return h + s


def d_trunk(P: Params, x, first_down: bool, n_res: int, train: bool):
    h = d_first_block(P, "main.0.", x, first_down, train)
    for k in range(1, n_res + 1):
        h = d_res_block(P, f"main.{k}.", h, train)
    return F.relu(h).sum(dim=(2, 3))


def _sn_linear(P, pre, x, train, bias=True):
    return F.linear(x, sn_weight(P, pre, train), P[pre + "bias"] if bias else None)


def image_discriminator(P: Params, x, train: bool = True):
    """models/discriminator.py:222-230."""
    f = d_trunk(P, x, True, 4, train)
    return _sn_linear(P, "classifier.", f, train, bias=False).view(-1)


def object_discriminator(P: Params, x, train: bool = True):
    """models/discriminator.py:264-278 (the `y` argument is ignored by the reference)."""
    f = d_trunk(P, x, False, 4, train)
    return _sn_linear(P, "classifier_src.", f, train).view(-1), _sn_linear(P, "classifier_cls.", f, train)


def attribute_discriminator(P: Params, x, train: bool = True, res128: bool = False):
    """models/discriminator.py:170-181 (64 px) / :130-141 (128 px variant, 5 residual blocks)."""
    f = d_trunk(P, x, False, 5 if res128 else 4, train)
    return _sn_linear(P, "classifier_att.", f, train)


# --------------------------------------------------------------------------- losses (train64.py)
def bce_logits(x, target_value: float):
    return F.binary_cross_entropy_with_logits(x, torch.full_like(x, target_value))


def kl_sum(mu, logvar):
    """train64.py:294-295 — sum, not mean."""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())
