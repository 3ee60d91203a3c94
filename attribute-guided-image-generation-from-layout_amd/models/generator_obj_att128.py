"""Drop-in for the reference's models/generator_obj_att128.py (128x128 generator: AdaptiveAvgPool2d(8)
in the layout encoder, nearest x2 + c5/spade_4/c6/spade_5/c7 decoder tail)."""
from . import _bootstrap  # noqa: F401
from agl.generator import (AttributeEncoder, ConditionalBatchNorm2d, ConvLSTMCell, CropEncoder, GlobalEncoder,  # noqa: F401
                           LayoutConvLSTM, ResidualBlock, get_z_random)
from agl.generator import Decoder as _Decoder, Generator as _Generator, LayoutEncoder as _LayoutEncoder
from .bilinear import crop_bbox_batch  # noqa: F401
from .spade.networks.normalization import SPADE  # noqa: F401


class LayoutEncoder(_LayoutEncoder):
    def __init__(self, conv_dim=64, z_dim=8, obj_att_dim=64, class_num=10, resi_num=6, clstm_layers=3, att_dim=64):
        super().__init__(conv_dim, z_dim, obj_att_dim, class_num, resi_num, clstm_layers, att_dim, pool_to_8=True)


class Decoder(_Decoder):
    def __init__(self, nf=64, conv_dim=64):
        super().__init__(nf, conv_dim, res128=True)


class Generator(_Generator):
    def __init__(self, num_embeddings, obj_att_dim=64, z_dim=8, obj_size=64, clstm_layers=3, attribute_dim=128):
        super().__init__(num_embeddings, obj_att_dim, z_dim, obj_size, clstm_layers, attribute_dim, res128=True)


Generator128 = Generator
