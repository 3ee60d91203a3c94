"""The SPADE-conditioned generator (64 px and 128 px variants) on the HIP kernels.

Module / attribute names follow the reference so `state_dict()` keys and `.parameters()` order match
(models/generator_obj_att.py:603-647 and models/generator_obj_att128.py:635-679); forward bodies are
sequences of agl.functional ops (fused conv / norm+ReLU / SPADE / ConvLSTM kernels).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import functional as F
from . import nn as A
from .convlstm import SequencePlan, layout_conv_lstm

import os
FRONT_STREAMS = True      # the rand / shift layout-encoder fronts on their branch streams (closed A/B, round 4)
# opt-in: the reconstruction branch's front + ConvLSTM on the third branch stream, so that its backward recurrence overlaps the batched
# rand / shift one.  Measured on one box, alternating (profiles/r04_ab_switches.txt): 466 / 472 images/s with it against 483 / 483 without
# at 64 px — two compute-bound recurrences beside each other run slower than one after the other — so the default keeps it off.
REC_CLSTM_STREAM = False


def get_z_random(batch_size, z_dim, random_type="gauss"):
    """Reference helper (generator_obj_att.py:10-15): drawn on the CPU generator, like the reference."""
    if random_type == "uni":
        return torch.rand(batch_size, z_dim) * 2.0 - 1.0
    return torch.randn(batch_size, z_dim)


class ConditionalBatchNorm2d(nn.Module):
    """generator_obj_att.py:31-44; statistics + per-object gamma/beta (+ReLU) in fused kernels."""

    def __init__(self, num_features, num_classes):
        super().__init__()
        self.num_features = num_features
        self.bn = A.BatchNorm2d(num_features, affine=False)
        self.embed = nn.Embedding(num_classes, num_features * 2)
        self.embed.weight.data[:, :num_features].normal_(1, 0.02)
        self.embed.weight.data[:, num_features:].zero_()

    def forward(self, x, y, relu=False):
        bn = self.bn
        return F.cond_batch_norm(x, self.embed.weight, y, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                 relu, self.training)


class ResidualBlock(nn.Module):
    """generator_obj_att.py:47-60: x + BN(conv3(ReLU(BN(conv3(x)))))."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.main = nn.Sequential(
            A.Conv2d(dim_in, dim_out, kernel_size=3, stride=1, padding=1, bias=False),
            A.BatchNorm2d(dim_out, affine=True, track_running_stats=True),
            nn.ReLU(inplace=True),
            A.Conv2d(dim_out, dim_out, kernel_size=3, stride=1, padding=1, bias=False),
            A.BatchNorm2d(dim_out, affine=True, track_running_stats=True))

    def forward(self, x):
        m = self.main
        t = m[1](m[0](x), relu=True)
        return m[4](m[3](t), residual=x)


class ConvLSTMCell(nn.Module):
    """Parameter holder for one ConvLSTM layer (generator_obj_att.py:63-118); the math runs in
    agl.convlstm over whole sequences."""

    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.height, self.width = input_size
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.bias = bias
        self.conv = A.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size=kernel_size, padding=self.padding, bias=bias)


class LayoutConvLSTM(nn.Module):
    """generator_obj_att.py:232-346."""

    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias=True, return_all_layers=False):
        super().__init__()
        hidden_dim = list(hidden_dim) if isinstance(hidden_dim, (list, tuple)) else [hidden_dim]
        if not (isinstance(kernel_size, tuple) or (isinstance(kernel_size, list) and all(isinstance(e, tuple) for e in kernel_size))):
            raise ValueError('`kernel_size` must be tuple or list of tuples')
        ks = kernel_size if isinstance(kernel_size, list) else [kernel_size] * len(hidden_dim)
        if len(ks) != len(hidden_dim):
            raise ValueError('Inconsistent list length.')
        assert all(k == (5, 5) for k in ks) and bias, "the HIP sequence kernel covers the 5x5, biased cells the path uses"
        self.height = self.width = input_size
        self.input_dim, self.hidden_dim, self.kernel_size = input_dim, hidden_dim, ks
        self.num_layers, self.bias, self.return_all_layers = len(hidden_dim), bias, return_all_layers
        self.cell_list = nn.ModuleList(
            [ConvLSTMCell((input_size, input_size), input_dim if i == 0 else hidden_dim[i - 1], hidden_dim[i], ks[i], bias)
             for i in range(self.num_layers)])

    def forward(self, obj_tensor, obj_to_img, hidden_state=None, plan: Optional[SequencePlan] = None):
        if hidden_state is not None:
            raise NotImplementedError()
        if plan is None:
            plan = SequencePlan(obj_to_img, obj_tensor.device)
        return layout_conv_lstm(obj_tensor, plan, self.hidden_dim, [c.conv.weight for c in self.cell_list],
                                [c.conv.bias for c in self.cell_list])


class CropEncoder(nn.Module):
    """generator_obj_att.py:367-422."""

    def __init__(self, conv_dim=64, z_dim=8, class_num=10):
        super().__init__()
        assert class_num > 0, "the path uses the class-conditional branch"
        d = conv_dim
        self.c1 = A.Conv2d(3, d, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn1 = ConditionalBatchNorm2d(d, class_num)
        self.c2 = A.Conv2d(d, d * 2, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn2 = ConditionalBatchNorm2d(d * 2, class_num)
        self.c3 = A.Conv2d(d * 2, d * 4, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn3 = ConditionalBatchNorm2d(d * 4, class_num)
        self.c4 = A.Conv2d(d * 4, d * 8, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn4 = ConditionalBatchNorm2d(d * 8, class_num)
        self.conv5 = A.Conv2d(d * 8, d * 16, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn5 = ConditionalBatchNorm2d(d * 16, class_num)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc_mu = A.Linear(d * 16, z_dim)
        self.fc_logvar = A.Linear(d * 16, z_dim)

    def forward(self, imgs, objs=None, eps=None):
        mu, logvar = self.trunk(imgs, objs)
        return self.sample(mu, logvar, eps), mu, logvar

    def trunk(self, imgs, objs=None):
        """Everything up to (mu, logvar): independent of the random draw."""
        # conv -> CondBN -> ReLU -> conv ...: every normalise-modulate(+ReLU) between two convolutions is folded into the staging pass
        # of the convolution that consumes it (F.norm_conv2d: the normalised tensor is never stored); the last one feeds the spatial mean
        x = self.c1(imgs)
        for bn, conv in ((self.bn1, self.c2), (self.bn2, self.c3), (self.bn3, self.c4), (self.bn4, self.conv5)):
            x = F.norm_conv2d(x, bn, objs, conv, relu=True, training=self.training)
        x = self.bn5(x, objs, relu=True)
        x = F.sum_hw(x, False, 1.0 / (x.shape[2] * x.shape[3]))
        return self.fc_mu(x), self.fc_logvar(x)

    @staticmethod
    def sample(mu, logvar, eps=None):
        if eps is None:
            eps = get_z_random(mu.size(0), mu.size(1))
        return F.reparameterize(mu, logvar, eps.to(mu.device))


class GlobalEncoder(nn.Module):
    """generator_obj_att.py:425-446."""

    def __init__(self):
        super().__init__()
        self.c1 = A.Conv2d(64, 128, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn1 = A.BatchNorm2d(128)
        self.c2 = A.Conv2d(128, 128, kernel_size=4, stride=2, padding=1, bias=False)

    def forward(self, h):
        return F.sum_hw(F.norm_conv2d(self.c1(h), self.bn1, None, self.c2, relu=True, training=self.training))


class LayoutEncoder(nn.Module):
    """generator_obj_att.py:449-513; the 128 px variant adds AdaptiveAvgPool2d(8) (generator_obj_att128.py:486,505)."""

    def __init__(self, conv_dim=64, z_dim=8, obj_att_dim=64, class_num=10, resi_num=6, clstm_layers=3, att_dim=64,
                 pool_to_8=False):
        super().__init__()
        hidden = {1: [64], 2: [64, 64], 3: [128, 64, 64]}[clstm_layers]
        self.clstm = LayoutConvLSTM(8, 512, hidden, (5, 5))
        self.residual = nn.Sequential(*[ResidualBlock(64, 64) for _ in range(resi_num)])
        d = conv_dim
        self.c0 = A.Conv2d(obj_att_dim + z_dim, d, kernel_size=1, stride=1, padding=1, bias=False)
        self.bn1 = ConditionalBatchNorm2d(d, class_num)
        self.c2 = A.Conv2d(d, d * 2, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn2 = ConditionalBatchNorm2d(d * 2, class_num)
        self.c3 = A.Conv2d(d * 2, d * 4, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn3 = ConditionalBatchNorm2d(d * 4, class_num)
        self.c4 = A.Conv2d(d * 4, d * 8, kernel_size=4, stride=2, padding=1, bias=False)
        self.bn4 = ConditionalBatchNorm2d(d * 8, class_num)
        if pool_to_8:
            self.pool = nn.AdaptiveAvgPool2d(8)
        self.pool_to_8 = pool_to_8
        self.closed_form_stage1 = True     # False: materialise c0's output and run bn1 / c2 as generic kernels

    def forward(self, objs_att, masks, obj_to_img, z, objs, plan: Optional[SequencePlan] = None):
        h = self.front(objs_att, masks, z, objs)
        h = self.clstm(h, obj_to_img, plan=plan)
        return self.residual(h)

    def forward_many(self, calls, obj_to_img, objs, residual=True, branch=None):
        """Several forward calls [(objs_att, masks, z), ...] on the same object list, results identical to calling
        forward() once per entry in order: the stages that carry batch statistics (CondBN before the ConvLSTM, BN in
        the residual blocks) run per call in call order, so every normalisation layer sees the same sequence of
        batches; the ConvLSTM between them has no batch coupling and runs ONCE on the concatenated object lists
        (k times the images per recurrence step: k times fewer launches, fuller grids, one backward).
        residual=False: stop before the residual blocks (the caller runs them per branch, possibly on separate streams)."""
        k = len(calls)
        if branch is None:
            fronts = [self.front(a, m, z, objs) for (a, m, z) in calls]
        else:
            # branch = (streams, private gradient arenas, deferred-update lists), one entry per call: the fronts are independent chains
            # of large-but-not-chip-filling kernels (per-object 4x4 / stride-2 convolutions and their ConditionalBatchNorms) and so are
            # their backward passes — the tail of the G step's backward, where nothing else runs.  Each front on its branch's stream,
            # with the branch's private gradient arena (the fronts share every parameter) and its BatchNorm updates deferred (applied
            # by the caller in call order, as in the sequential schedule).
            streams, arenas, deferred = branch
            main = torch.cuda.current_stream()
            # (the first stage updates bn1's running statistics inside its kernel: in call order on the caller's stream, private slots)
            stage1 = []
            for (a, m, z), ar in zip(calls, arenas):
                prev_a, F.GRAD_ARENA = F.GRAD_ARENA, ar
                try:
                    stage1.append(self.front_a(a, m, z, objs))
                finally:
                    F.GRAD_ARENA = prev_a
            fronts = []
            for h1, st, ar, lst in zip(stage1, streams, arenas, deferred):
                st.wait_stream(main)
                prev, prev_a = F.BN_DEFER, F.GRAD_ARENA
                F.BN_DEFER, F.GRAD_ARENA = lst, ar
                try:
                    with torch.cuda.stream(st):
                        fronts.append(self.front_b(h1, objs))
                finally:
                    F.BN_DEFER, F.GRAD_ARENA = prev, prev_a
                F.L.used_on(st, h1)
            for st, f in zip(streams, fronts):
                main.wait_stream(st)
                F.L.used_on(main, f)
        ids = obj_to_img.detach().cpu()
        n_img = int(ids.max()) + 1
        # image ids must stay one run per image: offset every copy past the previous one's last id
        plan = SequencePlan(torch.cat([ids + i * n_img for i in range(k)]), fronts[0].device)
        h = self.clstm(F.cat_batch(fronts), None, plan=plan)
        hs = F.split_batch(h, k)
        return list(hs) if not residual else [self.residual(hh) for hh in hs]

    def front(self, objs_att, masks, z, objs):
        return self.front_b(self.front_a(objs_att, masks, z, objs), objs)

    def front_a(self, objs_att, masks, z, objs):
        """c0 on the rank-1 input, bn1 + ReLU + c2 (closed form where the map allows): updates bn1's running statistics in place."""
        v = F.concat_channels(objs_att, z)
        assert self.c0.kernel_size == (1, 1) and self.c0.padding == (1, 1)
        u = F.linear(v, self.c0.weight.view(self.c0.out_channels, -1))   # c0 on the rank-1 tensor v (x) mask
        if self.closed_form_stage1 and masks.shape[-1] % 2 == 0 and self.c2.kernel_size == (4, 4):
            bn = self.bn1.bn                                              # bn1 + ReLU + c2 on two-level images
            h = F.layout_stage1(u, masks, objs, self.bn1.embed.weight, self.c2.weight, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, self.training)
        else:
            h = self.c2(self.bn1(F.mask_outer(u, masks, 1), objs, relu=True))
        return h

    def front_b(self, h, objs):
        """bn2 .. bn4 (+ the 128 px model's pool): every statistics call honours F.BN_DEFER."""
        h = F.norm_conv2d(h, self.bn2, objs, self.c3, relu=True, training=self.training)      # bn2 + ReLU folded into c3's staging pass
        h = F.norm_conv2d(h, self.bn3, objs, self.c4, relu=True, training=self.training)      # bn3 + ReLU into c4's
        h = self.bn4(h, objs)
        if self.pool_to_8:
            assert h.shape[2] == 16, "AdaptiveAvgPool2d(8) is an exact 2x2 mean on the 16x16 map of the 128 px model"
            h = F.avg_pool2(h)
        return h


class SPADE(nn.Module):
    """models/spade/networks/normalization.py:66-108 with param_free_norm = BatchNorm2d (:70,77-78).
    gamma and beta come from ONE 128->2C convolution whose output feeds the fused normalise-modulate(-ReLU)
    kernel; the nearest up-sampling of the 8x8 segmentation map is folded into the first conv's gather."""

    def __init__(self, norm_nc, label_nc):
        super().__init__()
        self.param_free_norm = A.BatchNorm2d(norm_nc, affine=False)
        nhidden = 128
        self.mlp_shared = nn.Sequential(A.Conv2d(label_nc, nhidden, kernel_size=3, padding=1), nn.ReLU())
        self.mlp_gamma = A.Conv2d(nhidden, norm_nc, kernel_size=3, padding=1)
        self.mlp_beta = A.Conv2d(nhidden, norm_nc, kernel_size=3, padding=1)

    block_grids = True     # False: both convolutions on the full-resolution grid (A/B tests)
    fold_gather = True     # False: write the expanded gamma|beta out before the modulation (A/B tests)

    def agl_param_pairs(self):
        """Parameters FlatParams should place back to back (their dim-0 concatenation becomes an arena view)."""
        return [("gb_weight", self.mlp_gamma.weight, self.mlp_beta.weight), ("gb_bias", self.mlp_gamma.bias, self.mlp_beta.bias)]

    def forward(self, x, segmap, relu=False, then=None):
        """then (optional): the ONE layer that reads the result (agl.nn.Conv2d / ConvTranspose2d); returns then(SPADE(x)) — as one graph
        node with the modulated tensor stored as bf16 inside it where bf16 arithmetic allows (F.spade_modulate_then)."""
        def modulate(gb, gather=None):
            n = self.param_free_norm
            if then is not None:
                return F.spade_modulate_then(x, gb, n.running_mean, n.running_var, n.num_batches_tracked, relu, self.training, gather, then)
            return F.spade_modulate(x, gb, n.running_mean, n.running_var, n.num_batches_tracked, relu, self.training, gather=gather)
        f = x.shape[2] // segmap.shape[2]
        up = f.bit_length() - 1
        assert segmap.shape[2] << up == x.shape[2] and segmap.shape[3] << up == x.shape[3], "power-of-two nearest up-sampling only"
        assert segmap.shape[2] == segmap.shape[3]
        wg, wb = self.mlp_gamma.weight, self.mlp_beta.weight
        joined = self.__dict__.get("_agl_joined")
        if joined is not None and joined["gb_weight"].data_ptr() == wg.data_ptr() and wg.requires_grad == joined["gb_weight"].requires_grad:
            # gamma and beta lie back to back in the flat arena (agl.flat.FlatParams): their concatenation is a view with its
            # own gradient slot — no concat kernels, no autograd adds
            w, b = joined["gb_weight"], joined["gb_bias"]
        else:
            w = F.concat_rows(wg, wb)
            b = F.concat_rows(self.mlp_gamma.bias, self.mlp_beta.bias)
            sg, sb = getattr(wg, "_agl_wsrc", None), getattr(wb, "_agl_wsrc", None)
            if sg is not None and sb is not None:      # packed-weight cache of the concatenation, valid while neither half changes
                w._agl_wsrc = F.L.WeightSrc(wg, (lambda: (sg.version(), sb.version())), tag="gamma|beta")
        nb = segmap.shape[2]
        if self.block_grids and f >= 4:
            # The f-fold nearest up-sampling of the segmentation map is constant on f x f blocks, so a 3x3 convolution of
            # it takes only 3 distinct values per block and axis (first row / interior / last row), and a 3x3
            # convolution of THAT only 5 (rows 0, 1, interior, f-2, f-1).  Both convolutions therefore run on class
            # grids — 3 cells per block for mlp_shared, 5 per block for gamma|beta when f >= 8 — and the results are
            # expanded by index maps; values are the ones the full-resolution convolutions produce (same taps, same
            # order), gradients are summed over each class (agl_grid_gather_bwd).  64 px: 24x24 / 40x40 instead of
            # 64x64 for SPADE_3; 128 px: 24x24 / 80x80 instead of 128x128.
            a3 = self.mlp_shared[0](F.grid_gather(segmap, "up3", nb), relu=True)
            if f >= 8:
                gb5 = F.conv2d(F.grid_gather(a3, "3to5", nb), w, b, 1, 1)
                if self.fold_gather:         # the 5-classes-per-block grid is expanded inside the modulation kernel's reads
                    return modulate(gb5, ("5tof", nb, f))
                gb = F.grid_gather(gb5, "5tof", nb, f)
            else:
                gb = F.conv2d(F.grid_gather(a3, "3tof", nb, f), w, b, 1, 1)
        else:
            # actv feeds exactly one convolution, which masks its input gradient by actv > 0 (no separate ReLU-backward pass)
            actv = self.mlp_shared[0](segmap, relu=True, up=up, relu_grad_by_consumer=True)
            gb = F.conv2d(actv, w, b, 1, 1, x_relu=True)
        return modulate(gb)


class Decoder(nn.Module):
    """generator_obj_att.py:516-572; 128 px tail generator_obj_att128.py:549-557, :587-604."""

    def __init__(self, nf=64, conv_dim=64, res128=False):
        super().__init__()
        self.sw, self.sh, self.h_dim = 8, 8, 64
        d = conv_dim
        self.c0_new = A.Conv2d(d + 128, d * 4, kernel_size=3, stride=1, padding=1, bias=False)
        self.spade_0 = SPADE(d * 4, self.h_dim)
        self.dc1 = A.ConvTranspose2d(d * 4, d * 4, kernel_size=4, stride=2, padding=1, bias=False)
        self.spade_1 = SPADE(d * 4, self.h_dim)
        self.dc2 = A.ConvTranspose2d(d * 4, d * 2, kernel_size=4, stride=2, padding=1, bias=False)
        self.spade_2 = SPADE(d * 2, self.h_dim)
        self.dc3 = A.ConvTranspose2d(d * 2, d, kernel_size=4, stride=2, padding=1, bias=False)
        self.spade_3 = SPADE(d, self.h_dim)
        self.c4 = A.Conv2d(d, 3, kernel_size=7, stride=1, padding=3, bias=True)
        self.res128 = res128
        if res128:
            self.c5 = A.Conv2d(3, d * 2, kernel_size=7, stride=1, padding=3, bias=False)
            self.spade_4 = SPADE(d * 2, self.h_dim)
            self.c6 = A.Conv2d(d * 2, d * 2, kernel_size=5, stride=1, padding=2, bias=False)
            self.spade_5 = SPADE(d * 2, self.h_dim)
            self.c7 = A.Conv2d(d * 2, 3, kernel_size=7, stride=1, padding=3, bias=True)

    def forward(self, hidden, global_h, z=None):
        seg = hidden
        h = self.c0_new(F.concat_channels(hidden, global_h))      # global vector broadcast over the 8x8 map
        # every SPADE output has ONE reader, the layer behind it: passed as `then`, the pair is one graph node and the modulated
        # tensor between them is stored as bf16 in bf16 arithmetic
        h = self.spade_0(h, seg, relu=True, then=self.dc1)
        h = self.spade_1(h, seg, relu=True, then=self.dc2)
        h = self.spade_2(h, seg, relu=True, then=self.dc3)
        h = self.spade_3(h, seg, relu=True, then=self.c4)
        if not self.res128:
            return h
        h = self.c5(h, up=1)                                   # nearest x2 folded into the 7x7 conv's gather
        h = self.spade_4(h, seg, relu=True, then=self.c6)
        return self.spade_5(h, seg, relu=True, then=self.c7)


class AttributeEncoder(nn.Module):
    """generator_obj_att.py:575-600."""

    def __init__(self, attribute_dim=106, embedding_dim=64, class_num=10):
        super().__init__()
        self.embedding = A.Embedding(class_num, embedding_dim)
        self.c0 = A.Linear(attribute_dim + embedding_dim, 128)
        self.bn0 = A.BatchNorm1d(128)
        self.c1 = A.Linear(128, 64)
        self.bn1 = A.BatchNorm1d(64)
        self.c2 = A.Linear(64, 64)

    def forward(self, objs, attribute):
        a = F.concat_channels(self.embedding(objs), attribute)
        a = self.bn0(self.c0(a), relu=True)
        a = self.bn1(self.c1(a), relu=True)
        return self.c2(a)


class Generator(nn.Module):
    """Drop-in for models.generator_obj_att.Generator / models.generator_obj_att128.Generator.

    forward(imgs, objs, boxes, masks, obj_to_img [CPU int64], z_rand, attribute, masks_shift, boxes_shift,
    attribute_est) -> the reference's 11-tuple.  `eps` (optional, three (O,z) tensors) pins the crop
    encoder's random draws; by default they are drawn on the CPU generator exactly like the reference.
    """

    def __init__(self, num_embeddings, obj_att_dim=64, z_dim=8, obj_size=64, clstm_layers=3, attribute_dim=128,
                 res128=False):
        super().__init__()
        self.obj_size = obj_size
        self.batch_clstm = True        # one ConvLSTM pass for the three layout-encoder calls (LayoutEncoder.forward_many)
        self.crop_encoder = CropEncoder(z_dim=z_dim, class_num=num_embeddings)
        self.layout_encoder = LayoutEncoder(z_dim=z_dim, obj_att_dim=obj_att_dim, class_num=num_embeddings,
                                            clstm_layers=clstm_layers, pool_to_8=res128)
        self.decoder = Decoder(res128=res128)
        self.global_encoder = GlobalEncoder()
        self.attribute_encoder = AttributeEncoder(attribute_dim=attribute_dim, embedding_dim=obj_att_dim,
                                                  class_num=num_embeddings)

    @staticmethod
    def _with_conv_stats(fn):
        """Convolutions of the generator leave BatchNorm partial sums for the norm that follows them (F.EMIT_STATS)."""
        def run(self, *a, **k):
            prev, F.EMIT_STATS = F.EMIT_STATS, True
            try:
                return fn(self, *a, **k)
            finally:
                F.EMIT_STATS = prev
        run.__name__, run.__doc__ = fn.__name__, fn.__doc__
        return run

    def forward(self, imgs, objs, boxes, masks, obj_to_img, z_rand, attribute, masks_shift, boxes_shift, attribute_est,
                eps: Optional[Sequence[torch.Tensor]] = None):
        sh = self.part_a(imgs, objs, boxes, masks, obj_to_img, z_rand, attribute, masks_shift, boxes_shift, attribute_est)
        e = self.draw_eps(sh, eps)
        rec = self.part_rec(sh, e[0])
        self.part_b(sh)
        return self.outputs(sh, rec, e)

    # The forward pass in four parts.  Only part_rec and the two re-parameterisations of `outputs` depend on the random
    # draws eps; part_a and part_b depend on the inputs and the weights alone.  A training iteration evaluates the
    # generator twice with unchanged weights (train64.py:195 and :280), so agl.trainer evaluates part_a / part_b once,
    # keeps their graph, and runs part_rec twice.  Per BatchNorm layer the order of batches (rec, rand, shift) is the
    # reference's in either schedule.
    @_with_conv_stats.__func__
    def part_a(self, imgs, objs, boxes, masks, obj_to_img, z_rand, attribute, masks_shift, boxes_shift, attribute_est):
        A._need_device(imgs)
        dev = imgs.device
        sh = dict(imgs=imgs, objs=objs, boxes=boxes, masks=masks, obj_to_img=obj_to_img, z_rand=z_rand,
                  masks_shift=masks_shift, boxes_shift=boxes_shift, o2i_dev=F.L.box_map_to_device(obj_to_img, dev),
                  plan=SequencePlan(obj_to_img, dev))
        sh["crops_input"] = F.crop_boxes(imgs, boxes, sh["o2i_dev"], self.obj_size)
        sh["mu"], sh["logvar"] = self.crop_encoder.trunk(sh["crops_input"], objs)
        sh["objs_att"] = self.attribute_encoder(objs, attribute)
        ev = getattr(attribute_est, "_agl_ready", None)      # produced on another stream (agl.trainer's pre-step): wait for it here
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        sh["objs_att_est"] = self.attribute_encoder(objs, attribute_est)
        return sh

    @staticmethod
    def draw_eps(sh, eps=None):
        """The three crop-encoder draws of one forward pass, in the reference's order (CPU generator)."""
        if eps is not None:
            return list(eps)
        O, zd = sh["mu"].shape
        return [get_z_random(O, zd) for _ in range(3)]

    @_with_conv_stats.__func__
    def part_rec(self, sh, eps0):
        z_rec = self.crop_encoder.sample(sh["mu"], sh["logvar"], eps0)
        streams, arenas = self.__dict__.get("branch_streams"), self.__dict__.get("branch_grad_arenas")
        if REC_CLSTM_STREAM and streams is not None and len(streams) >= 3 and arenas is not None and torch.is_grad_enabled() and self.training:
            # The reconstruction branch's layout-encoder front and ConvLSTM on the third branch stream (private gradient arena): the
            # forward order is unchanged (the caller's stream waits for the result), but autograd runs their BACKWARD on that stream —
            # the recurrence's chain of small per-step kernels then overlaps the batched rand / shift recurrence's chain instead of
            # queueing behind it in the tail of the G step's backward, where nothing else runs.
            main, g2 = torch.cuda.current_stream(), streams[2]
            g2.wait_stream(main)
            prev_a, F.GRAD_ARENA = F.GRAD_ARENA, arenas[2]
            try:
                with torch.cuda.stream(g2):
                    le = self.layout_encoder
                    h0 = le.clstm(le.front(sh["objs_att_est"], sh["masks"], z_rec, sh["objs"]), sh["obj_to_img"], plan=sh["plan"])
            finally:
                F.GRAD_ARENA = prev_a
            main.wait_stream(g2)
            F.L.used_on(main, h0)
            h_rec = self.layout_encoder.residual(h0)
        else:
            h_rec = self.layout_encoder(sh["objs_att_est"], sh["masks"], sh["obj_to_img"], z_rec, sh["objs"], sh["plan"])
        img_rec = self.decoder(h_rec, self.global_encoder(h_rec))
        crops_input_rec = F.crop_boxes(img_rec, sh["boxes"], sh["o2i_dev"], self.obj_size)
        return img_rec, crops_input_rec

    @_with_conv_stats.__func__
    def part_rec_nograd_beside_b(self, sh, eps0, tape_b):
        """part_rec WITHOUT a graph (the D step's reconstruction branch, train64.py:195) evaluated beside part_b: its layout-encoder
        front runs first on the caller's stream (the CondBN layers of the fronts update their running statistics in place, and the
        reference's order is rec, rand, shift); its ConvLSTM, residual blocks, decoder and crop then run on the third branch stream
        while part_b proceeds, with their BatchNorm updates deferred and applied before part_b's (rec, rand, shift again).
        tape_b: the BatchNorm tape part_b's layers are recorded on.  Returns part_rec's outputs."""
        streams = self.__dict__["branch_streams"]
        main, g2 = torch.cuda.current_stream(), streams[2]
        with torch.no_grad():
            z_rec = self.crop_encoder.sample(sh["mu"], sh["logvar"], eps0)
            f_rec = self.layout_encoder.front(sh["objs_att_est"], sh["masks"], z_rec, sh["objs"])
            g2.wait_stream(main)
            lst, prev = [], F.BN_DEFER
            F.BN_DEFER = lst
            try:
                with torch.cuda.stream(g2):
                    h = self.layout_encoder.residual(self.layout_encoder.clstm(f_rec, sh["obj_to_img"], plan=sh["plan"]))
                    img_rec = self.decoder(h, self.global_encoder(h))
                    crops_rec = F.crop_boxes(img_rec, sh["boxes"], sh["o2i_dev"], self.obj_size)
            finally:
                F.BN_DEFER = prev
        F.BN_TAPE = tape_b
        try:
            self.part_b(sh, before_deferred=lambda: (main.wait_stream(g2), F.L.used_on(main, img_rec, crops_rec), F.bn_apply_deferred(lst)))
        finally:
            F.BN_TAPE = None
        return img_rec, crops_rec

    @_with_conv_stats.__func__
    def part_b(self, sh, before_deferred=None):
        objs, o2i = sh["objs"], sh["obj_to_img"]
        calls = [(sh["objs_att"], sh["masks"], sh["z_rand"]), (sh["objs_att"], sh["masks_shift"], sh["z_rand"])]
        streams = self.__dict__.get("branch_streams")
        s = self.obj_size
        if streams is not None and self.batch_clstm and self.training:
            # The `rand` and `shift` branches are independent after the (batched) ConvLSTM: residual blocks, global encoder,
            # decoder, crop and crop-encoder trunk of each run on a stream of their own (their small grids — 64 workgroups in the
            # residual blocks — and partial last rounds overlap), and so does their backward, which autograd runs on the forward's
            # streams.  The BatchNorm layers they share update their running statistics afterwards, rand first, then shift —
            # the order of the sequential schedule (F.BN_DEFER).
            arenas = self.__dict__.get("branch_grad_arenas") or [None, None]      # private gradient slots per branch (F.GRAD_ARENA)
            deferred = [[], []]
            fronts_on_streams = FRONT_STREAMS and arenas[0] is not None
            hs = self.layout_encoder.forward_many(calls, o2i, objs, residual=False,
                                                  branch=(streams[:2], arenas[:2], deferred) if fronts_on_streams else None)
            main = torch.cuda.current_stream()
            for k, (st, h0, tag, boxes) in enumerate(((streams[0], hs[0], "rand", sh["boxes"]), (streams[1], hs[1], "shift", sh["boxes_shift"]))):
                st.wait_stream(main)
                lst, prev, prev_a = deferred[k], F.BN_DEFER, F.GRAD_ARENA
                F.BN_DEFER, F.GRAD_ARENA = lst, arenas[k]
                try:
                    with torch.cuda.stream(st):
                        h = self.layout_encoder.residual(h0)
                        img = self.decoder(h, self.global_encoder(h))
                        crops = F.crop_boxes(img, boxes, sh["o2i_dev"], s)
                        mu, lv = self.crop_encoder.trunk(crops, objs)
                finally:
                    F.BN_DEFER, F.GRAD_ARENA = prev, prev_a
                sh["img_" + tag], sh["crops_" + tag], sh["mu_" + tag], sh["lv_" + tag] = img, crops, mu, lv
            for st in streams[:2]:
                main.wait_stream(st)
            for tag in ("rand", "shift"):      # produced on the branch streams, read on the caller's (and the discriminators') from here on
                F.L.used_on(main, sh["img_" + tag], sh["crops_" + tag], sh["mu_" + tag], sh["lv_" + tag])
            if before_deferred is not None:
                before_deferred()
            for lst in deferred:
                F.bn_apply_deferred(lst)
            return sh
        assert before_deferred is None, "part_rec_nograd_beside_b needs the concurrent schedule"
        if self.batch_clstm:
            h_rand, h_shift = self.layout_encoder.forward_many(calls, o2i, objs)
        else:
            h_rand, h_shift = [self.layout_encoder(a, m, o2i, z, objs, sh["plan"]) for (a, m, z) in calls]
        g_rand = self.global_encoder(h_rand)
        g_shift = self.global_encoder(h_shift)
        sh["img_rand"] = self.decoder(h_rand, g_rand)
        sh["img_shift"] = self.decoder(h_shift, g_shift)
        sh["crops_rand"] = F.crop_boxes(sh["img_rand"], sh["boxes"], sh["o2i_dev"], s)
        sh["mu_rand"], sh["lv_rand"] = self.crop_encoder.trunk(sh["crops_rand"], objs)
        sh["crops_shift"] = F.crop_boxes(sh["img_shift"], sh["boxes_shift"], sh["o2i_dev"], s)
        sh["mu_shift"], sh["lv_shift"] = self.crop_encoder.trunk(sh["crops_shift"], objs)
        return sh

    def outputs(self, sh, rec, eps):
        # The reference keeps the SECOND value the crop encoder returns (`_, z_rand_rec, _ = self.crop_encoder(...)`,
        # generator_obj_att.py:639,644), i.e. mu — the samples drawn with eps[1], eps[2] are discarded there (the draws
        # are still consumed from the CPU generator, which draw_eps reproduces).
        img_rec, crops_input_rec = rec
        return (sh["crops_input"], crops_input_rec, sh["crops_rand"], sh["crops_shift"], img_rec, sh["img_rand"],
                sh["img_shift"], sh["mu"], sh["logvar"], sh["mu_rand"], sh["mu_shift"])
